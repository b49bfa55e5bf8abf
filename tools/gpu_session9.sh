export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ksums or cfg3" 2>&1 | tail -4
for v in shipped prefetch shipped prefetch; do
if [ $v = shipped ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --workload cfg3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/nu_$v.json 2> $O/nu_$v.err
python - <<PY
import json
d=json.loads(open('$O/nu_$v.json').read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], d['value'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})
PY
done
