R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -k "sample or design or cfg1 or model" 2>&1 | tail -3
for rep in 1 2; do
for side in 1 0; do
NAMP_ORDER_SIDE=$side timeout 600 python bench.py --workload cfg1 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side $side cfg1', d['ms_per_step'], d.get('latency_ms'))"
done
done
NAMP_ORDER_SIDE=1 timeout 600 python bench.py --workload cfg1s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side 1 cfg1s', d['ms_per_step'], d.get('latency_ms'))"
NAMP_ORDER_SIDE=0 timeout 600 python bench.py --workload cfg1s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side 0 cfg1s', d['ms_per_step'], d.get('latency_ms'))"
