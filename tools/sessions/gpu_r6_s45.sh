R=$GRAFT_REPO_ROOT; cd $R
NAMP_LIB_PATH=$R/tools/_variants/feat_stamps.so timeout 600 python tools/feat_stamps.py 2>&1 | grep -v amdgpu.ids | head -24
