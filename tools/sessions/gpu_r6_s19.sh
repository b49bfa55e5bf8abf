R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/cfg3_ab.py --masks 11,15 --reps 2 2>&1 | grep mask
NAMP_LIB_PATH=$R/tools/_variants/stamps.so timeout 300 python tools/p32_stamps.py 2>&1 | grep -A12 "embedding"
