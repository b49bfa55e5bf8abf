R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
echo "== shipped"; timeout 600 python tools/score_ab.py 2>&1 | grep "side_stream=True" | tail -3
for v in x3vprio1 x3vprio3; do
echo "== $v"; NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 600 python tools/score_ab.py 2>&1 | grep "side_stream=True" | tail -3
done
done
