set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent" > $O/pytest_persist.log 2>&1; echo "rc=$?" >> $O/pytest_persist.log
tail -5 $O/pytest_persist.log
for p in fp32 x3; do
NAMP_PERSISTENT=0 timeout 300 python bench.py --steps 200 --warmup 10 --precision $p --no-cpu-baseline --no-gather --no-secondary > $O/bench_chain_$p.json 2> $O/bench_chain_$p.err
NAMP_PERSISTENT=1 timeout 300 python bench.py --steps 200 --warmup 10 --precision $p --no-cpu-baseline --no-gather --no-secondary > $O/bench_persist_$p.json 2> $O/bench_persist_$p.err
for v in $(ls tools/_variants/*.so 2>/dev/null); do
n=$(basename $v .so)
NAMP_LIB_PATH=$R/$v NAMP_PERSISTENT=1 timeout 300 python bench.py --steps 200 --warmup 10 --precision $p --no-cpu-baseline --no-gather --no-secondary > $O/bench_${n}_$p.json 2> $O/bench_${n}_$p.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['per_kernel'].items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
