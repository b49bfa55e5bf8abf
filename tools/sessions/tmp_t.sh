cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "reduce_sum or golden or on_chip" 2>&1 | tail -3
timeout 600 python tools/train_time.py --steps 3 --profile --each reduce_sum 2>&1 | grep -E "ms/step|== reduce_sum" | cut -c1-600
timeout 600 python tools/train_time.py --steps 5 --precision bf16 2>&1 | grep -E "ms/step"
