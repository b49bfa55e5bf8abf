R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_bench.py -x -q -k "two_ranks" 2>&1 | tail -60
