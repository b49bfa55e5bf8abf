R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "feature_weight or golden or mixed" 2>&1 | tail -3
for nw in 8 4; do for prec in bf16 x3; do
  echo "== NAMP_FEATW_WAVES=$nw $prec"
  NAMP_FEATW_WAVES=$nw timeout 600 python tools/train_time.py --steps 4 --precision $prec --profile 2>&1 | grep -E "ms/step|feat_wgrad_x3_kernel|reduce_sum" | cut -c1-150
done; done
