# Round 4, session 12: FusedAdam without the per-step synchronisation — cfg5 step time and the optimiser tests
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "adam or golden or reduces_loss or checkpoint or mixed" 2>&1 | tail -3
python tools/train_time.py --precision bf16 --steps 10 2>&1 | grep -v amdgpu.ids | tail -2
python tools/train_time.py --steps 10 2>&1 | grep -v amdgpu.ids | tail -2
