R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/sample_profile.py 2>&1 | grep "dec_sample_kernel\|host enqueue" | cut -c1-150
NAMP_LIB_PATH=$R/tools/_variants/wstamps.so timeout 300 python tools/sample_wstamps.py 2>&1 | grep -v amdgpu.ids | head -40
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "sample or sampl" 2>&1 | tail -4
