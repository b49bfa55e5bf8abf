export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg3 or bf16" 2>&1 | tail -4
for v in 16 32 16 32; do
NAMP_BF16S_TABLES=$v timeout 600 python bench.py --workload cfg3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/tbl_$v.json 2> $O/tbl_$v.err
python - <<PY
import json
d=json.loads(open('$O/tbl_$v.json').read().strip().splitlines()[-1])
print($v, d['ms_per_step'], d['value'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})
PY
done
