cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "autograd or golden or odd_shapes" 2>&1 | tail -3
timeout 600 python tools/train_time.py --steps 5 --precision bf16 --profile 2>&1 | grep -E "ms/step|scatter_rows" | cut -c1-100
timeout 600 python tools/train_time.py --steps 5 --profile 2>&1 | grep -E "ms/step|scatter_rows" | cut -c1-100
