# Round 4, session 5: stationary backward with a round of look-ahead on every HBM stream
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1; do python tools/dw_time.py --prec 2 --mode $m; done
python tools/dw_time.py --prec 2 --mode 0 --old
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "edge_mlp_backward" 2>&1 | tail -2
