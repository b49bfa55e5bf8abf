# Round 3, session 12: what would plain-bf16 residue-level GEMMs buy the bf16 mode (ablation NAMP_ABL_X1: hi.hi product only)?
R=$GRAFT_REPO_ROOT
cd $R
for v in base x1 base x1; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})"
done
export NAMP_LIB_PATH=$R/tools/_variants/x1.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16_throughput or cfg3" -s 2>&1 | grep -v amdgpu | tail -8
