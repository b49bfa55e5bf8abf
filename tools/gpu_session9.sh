export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
for v in head new head new; do
if [ $v = new ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --workload cfg4 --no-cpu-baseline > $O/rbf_$v.json 2> $O/rbf_$v.err
python -c "
import json
d=json.loads(open('$O/rbf_$v.json').read().strip().splitlines()[-1]); print('$v cfg4', d['value'], d['ms_per_step'])"
done
