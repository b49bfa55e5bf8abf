R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "class_sums or slices or reduce_sum" 2>&1 | tail -15
