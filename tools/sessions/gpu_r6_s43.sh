R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for w in 12 16; do
echo "== waves $w"; NAMP_FEAT_WAVES=$w timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0'
done
done
NAMP_FEAT_WAVES=16 timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
