R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
for f in 1 0; do
NAMP_ORDER_FOLD=$f timeout 600 python tools/score_ab.py 2>&1 | grep "cfg2 side_stream=True" | tail -2 | sed "s/^/fold $f /"
done
done
