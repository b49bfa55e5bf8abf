R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s5
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -6
timeout 300 python tools/sample_time.py 2>&1 | grep "sample()"
