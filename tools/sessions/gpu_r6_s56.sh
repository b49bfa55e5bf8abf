R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -4
FEAT_MASKS=11,43,299 timeout 600 python tools/feat_parts_ab.py 2>&1 | grep cfg2
for m in 43 299; do
NAMP_BF16P=$m timeout 600 python tools/score_ab.py 2>&1 | grep "cfg2 side_stream=True" | sed "s/^/mask $m /"
done
for m in 43 299; do
NAMP_BF16P=$m timeout 600 python tools/score_ab.py 2>&1 | grep "cfg2 side_stream=True" | sed "s/^/mask $m /"
done
