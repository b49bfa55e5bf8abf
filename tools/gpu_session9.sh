export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
for v in head new head new; do
if [ $v = new ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --workload cfg5 --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline > $O/c5b_$v.json 2> $O/c5b_$v.err
python -c "
import json
e=json.loads(open('$O/c5b_$v.json').read().strip().splitlines()[-1]); print('$v cfg5 mixed', e['ms_per_step'], {k[:48]:v for k,v in e['per_kernel_ms_per_step'].items() if 'bf16_persistent' in k})"
done
