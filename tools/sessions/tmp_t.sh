cd $GRAFT_REPO_ROOT; timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
timeout 600 python tools/train_time.py --steps 5 --precision bf16 --profile --shapes 2>&1 | grep -E "ms/step|== device|aten::" | head -12 | cut -c1-150
timeout 600 python tools/train_time.py --steps 5 2>&1 | grep -E "ms/step"
