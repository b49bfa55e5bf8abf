export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -8
for v in 1 0 1 0; do
NAMP_TAIL_VALU=$v timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 200 > $O/tailx3_$v.json 2> $O/tailx3_$v.err
python - <<PY
import json
d=json.loads(open('$O/tailx3_$v.json').read().strip().splitlines()[-1])
print('VALU=$v', d['ms_per_step'], 'x3', d['x3']['ms_per_step'], d['x3'].get('parity_vs_cpu',{}).get('max_abs_dlogp_vs_cpu'), {k:v['avg_ms'] for k,v in d['x3']['per_kernel'].items()})
PY
done
