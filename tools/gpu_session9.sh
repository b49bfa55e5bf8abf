export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3
for v in head new head new; do
if [ $v = new ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 200 > $O/fc_$v.json 2> $O/fc_$v.err
python - <<PY
import json
d=json.loads(open('$O/fc_$v.json').read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], 'x3', d['x3']['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()}, {k:v['avg_ms'] for k,v in d['x3']['per_kernel'].items()})
PY
done
