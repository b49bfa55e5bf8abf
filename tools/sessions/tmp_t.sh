cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "positional or feature_weight or golden" 2>&1 | tail -5
timeout 600 python tools/train_time.py --steps 5 --precision bf16 --profile 2>&1 | grep -E "ms/step|pos_grad" | cut -c1-120
