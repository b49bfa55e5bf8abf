R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
NAMP_LIB_PATH=$R/tools/_variants/feat_stamps.so timeout 600 python tools/feat_stamps.py 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/feat_parts_ab.py 2>&1 | tail -16
