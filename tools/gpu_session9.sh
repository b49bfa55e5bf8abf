export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
for v in head new head new; do
if [ $v = new ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --workload cfg4 --no-cpu-baseline > $O/c4_$v.json 2> $O/c4_$v.err
timeout 600 python bench.py --workload cfg5 --precision x3 --steps 8 --warmup 3 --no-cpu-baseline > $O/c5_$v.json 2> $O/c5_$v.err
python -c "
import json
d=json.loads(open('$O/c4_$v.json').read().strip().splitlines()[-1]); e=json.loads(open('$O/c5_$v.json').read().strip().splitlines()[-1]); print('$v cfg4', d['value'], d['ms_per_step'], 'cfg5', e['ms_per_step'])"
done
