R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['x3']['ms_per_step'], 'fullfwd', d['cpu_baseline'].get('gpu_full_forward_ms'))"
timeout 600 python bench.py --workload cfg1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1', d['ms_per_step'], d.get('latency_ms'))"
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'])"
