R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
echo "== shipped"; timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0' | cut -c1-42
for v in fe_prio1 fe_prio3; do
echo "== $v"; NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0' | cut -c1-42
done
done
