R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/sample_profile.py 2>&1 | grep -v amdgpu.ids | head -14 | cut -c1-150
timeout 2000 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -4
python bench.py --workload cfg1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1', d['ms_per_step'], d.get('levels'))"
