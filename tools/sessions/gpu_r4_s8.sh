# Round 4, session 8: x3 ring kernel after the branch-free / LDS-barrier changes
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1; do python tools/dw_time.py --prec 1 --mode $m; done
python tools/dw_time.py --prec 1 --mode 0 --old

timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "edge_mlp_backward or golden or odd_shapes" 2>&1 | tail -3
