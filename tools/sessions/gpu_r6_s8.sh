R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "node_update_w" 2>&1 | grep "node_update_w vs\|passed\|failed"
timeout 600 python tools/cfg3_ab.py --masks 3,11 --reps 2 2>&1 | grep mask
