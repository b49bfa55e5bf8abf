export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
for v in head new head new; do
if [ $v = new ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
timeout 600 python bench.py --workload cfg3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/pa_$v.json 2> $O/pa_$v.err
python -c "
import json
d=json.loads(open('$O/pa_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})"
done
