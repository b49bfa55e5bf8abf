R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
timeout 600 python tools/cfg3_ab.py --masks 11 --reps 2 2>&1 | grep mask
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'])"
