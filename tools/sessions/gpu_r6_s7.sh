R=$GRAFT_REPO_ROOT; cd $R
for v in nw_nogemm nw_nostage nw_nogelu nw_noin nw_skel; do NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 300 python tools/cfg3_ab.py --masks 11 --reps 1 2>&1 | grep mask; done
