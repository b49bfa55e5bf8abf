# Round 4, session 13: W_e and the hoisted-table products at the step's precision
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -q 2>&1 | tail -8
python tools/train_time.py --precision bf16 --steps 10 2>&1 | grep -v amdgpu.ids | tail -1
python tools/train_time.py --steps 10 2>&1 | grep -v amdgpu.ids | tail -1
