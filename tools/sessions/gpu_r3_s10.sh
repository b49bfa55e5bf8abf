R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s10
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "symmetric or sample" > $O/sym.log 2>&1; tail -8 $O/sym.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg3 or bf16" -s > $O/bf16.log 2>&1; tail -8 $O/bf16.log
timeout 300 python tools/sample_sym_time.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})"; done
