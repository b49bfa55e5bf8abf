R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -s 2>&1 | grep "featuriser in\|passed\|failed\|Error" | tail -40
timeout 600 python tools/feat_parts_ab.py 2>&1 | tail -20
for m in 11 43 107 171; do
NAMP_BF16P=$m timeout 600 python tools/score_ab.py 2>&1 | grep "cfg2 side_stream=True" | sed "s/^/mask $m /"
done
