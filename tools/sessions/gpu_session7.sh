export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -6
for p in x3 bf16; do
timeout 600 python bench.py --workload cfg5 --precision $p --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_$p.json 2> $O/bench_cfg5_$p.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02h/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d.get('hip_kernel_share'), d['whole_step'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
