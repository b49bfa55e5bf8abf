R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/sample_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-160
NAMP_LIB_PATH=$R/tools/_variants/stamps.so timeout 300 python tools/sample_stamps.py 2>&1 | grep -v amdgpu.ids
