export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --workload cfg3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/l2_shipped.json 2> $O/l2_shipped.err
NAMP_NODE_T4=1 timeout 600 python bench.py --workload cfg3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/l2_t4.json 2> $O/l2_t4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02k/l2_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
