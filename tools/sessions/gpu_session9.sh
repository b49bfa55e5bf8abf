export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 2000 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
for v in head new head new; do
if [ $v = new ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --workload cfg5 --precision x3 --steps 8 --warmup 3 --no-cpu-baseline > $O/bw_$v.json 2> $O/bw_$v.err
python -c "
import json
e=json.loads(open('$O/bw_$v.json').read().strip().splitlines()[-1]); print('$v cfg5', e['ms_per_step'], {k[:40]:v for k,v in e['per_kernel_ms_per_step'].items() if 'chain_bwd' in k})"
done
