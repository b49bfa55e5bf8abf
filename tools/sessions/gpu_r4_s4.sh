# Round 4, session 4: where the weight-stationary backward's round goes — ablation builds (DW_EXP_*) at the cfg5 size
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1; do
  python tools/dw_time.py --prec 2 --mode $m --old
  python tools/dw_time.py --prec 2 --mode $m
done
python tools/dw_time.py --prec 1 --mode 0 --old
python tools/dw_time.py --prec 1 --mode 0
for v in nocontract nostage nogemm nogelu nostore noload chainonly; do
  NAMP_LIB_PATH=$R/tools/_variants/$v.so python tools/dw_time.py --prec 2 --mode 0
done
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "edge_mlp_backward" 2>&1 | tail -2
