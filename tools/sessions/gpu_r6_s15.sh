R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "two_tile or headline_size or padded_batch or knn_selection or maximum_size" 2>&1 | tail -4
timeout 600 python tools/feat_ab.py 2>&1 | grep mask
