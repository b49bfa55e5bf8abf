R=$GRAFT_REPO_ROOT; cd $R
echo "== shipped"; timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0' | cut -c1-40
for v in fa_nomul fa_nogen fa_nodma fa_nobar fa_nomul_nogen fa_skel; do
echo "== $v"; NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0' | cut -c1-40
done
