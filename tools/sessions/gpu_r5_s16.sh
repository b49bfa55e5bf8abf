R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s16
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-gather > gpurun_out/r5s16/bench_cfg2_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5s16/bench_cfg2_$i.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['ms_per_step'], d['x3']['ms_per_step'], 'traffic', r.get('traffic'), r.get('traffic_algorithmic'), 'launch', r['avg_launch_ms'], 'x3 traffic', d['x3'].get('traffic'))
PY
done
