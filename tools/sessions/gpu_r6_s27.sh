R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -s 2>&1 | grep "two-part\|passed\|failed\|Error" | tail -12
for m in 11 43; do
NAMP_BF16P=$m timeout 600 python tools/score_ab.py 2>&1 | grep "side_stream=True" | sed "s/^/mask $m /"
NAMP_BF16P=$m timeout 600 python bench.py --workload cfg1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $m cfg1', d['ms_per_step'], d.get('latency_ms'))"
done
