R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench.py -x -q 2>&1 | tail -6
timeout 600 python bench.py --workload cfg1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1', d['ms_per_step'], d.get('latency_ms'), d.get('latency_ms_min'))"
timeout 600 python bench.py --workload cfg1 --specificity --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1s', d['ms_per_step'], d.get('latency_ms'), d.get('latency_ms_min'))"
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'])"
timeout 600 python tools/cfg3_ab.py --masks 3,11 --reps 2 2>&1 | grep mask
