R=$GRAFT_REPO_ROOT; cd $R
for v in st_skel st_skel_nolds; do echo "#### $v"; NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 300 python tools/p32_stamps.py 2>&1 | grep -v Warning | grep -A12 "edge update\|dec message"; done
for v in p_noldsw p_nomfma p_nomem p_nogelu; do NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 300 python tools/cfg3_ab.py --masks 3 --reps 1 2>&1 | grep mask; done
