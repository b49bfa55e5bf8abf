R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s3
timeout 600 python tools/train_time.py --steps 6 --precision bf16 --profile > gpurun_out/r5s3/train_bf16.txt 2>&1
timeout 600 python tools/train_time.py --steps 6 --profile > gpurun_out/r5s3/train_x3.txt 2>&1
grep -h "ms/step" gpurun_out/r5s3/*.txt
