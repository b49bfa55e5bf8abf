R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for v in feat_nopipe fp_v1 fp_v4 fp_v5 fp_v6; do
echo "== $v"; NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0'
done
done
