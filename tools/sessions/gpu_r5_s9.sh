R=$GRAFT_REPO_ROOT; cd $R
NAMP_LIB_PATH=$R/tools/_variants/wstamps.so timeout 300 python tools/sample_wstamps.py 2>&1 | grep -v amdgpu.ids | head -40
