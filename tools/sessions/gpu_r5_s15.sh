R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s15
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
timeout 600 python tools/train_time.py --steps 6 --precision bf16 --profile > gpurun_out/r5s15/train_bf16.txt 2>&1
timeout 600 python tools/train_time.py --steps 6 --profile > gpurun_out/r5s15/train_x3.txt 2>&1
grep -h "ms/step\|== device" gpurun_out/r5s15/*.txt
