export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg3 or bf16 or wide or random_shapes" 2>&1 | tail -3
for v in 1 2; do
timeout 600 python bench.py --workload cfg3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/cst_$v.json 2> $O/cst_$v.err
python -c "
import json
d=json.loads(open('$O/cst_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})"
done
