R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
echo "== pipelined"; timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu
echo "== not pipelined"; NAMP_LIB_PATH=$R/tools/_variants/feat_nopipe.so timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
