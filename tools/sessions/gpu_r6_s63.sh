R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python tools/score_noorder.py 2>&1 | grep -v amdgpu
