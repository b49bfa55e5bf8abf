R=$GRAFT_REPO_ROOT; cd $R
NAMP_LIB_PATH=$R/tools/_variants/stamps.so timeout 300 python tools/sample_stamps.py 2>&1 | grep -v amdgpu.ids | head -20
