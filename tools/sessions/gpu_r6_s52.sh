R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
echo "== early W_e"; timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0'
echo "== before"; NAMP_LIB_PATH=$R/tools/_variants/fe_noearly.so timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0'
done
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py tests/test_gpu_train.py -x -q 2>&1 | tail -3
