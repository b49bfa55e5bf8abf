R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for f in 1 0; do
NAMP_ORDER_FOLD=$f timeout 600 python tools/score_ab.py 2>&1 | grep "side_stream=True" | tail -2 | sed "s/^/fold $f /"
NAMP_ORDER_FOLD=$f timeout 600 python bench.py --workload cfg1 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold $f cfg1', d['ms_per_step'], d.get('latency_ms'))"
done
done
