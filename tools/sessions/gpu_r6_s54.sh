R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
echo "== shipped"; timeout 600 python tools/score_ab.py 2>&1 | grep "side_stream=True" | tail -4
echo "== x3prio1"; NAMP_LIB_PATH=$R/tools/_variants/x3prio1.so timeout 600 python tools/score_ab.py 2>&1 | grep "side_stream=True" | tail -4
done
