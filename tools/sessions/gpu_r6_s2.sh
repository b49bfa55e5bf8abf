# round 6, session 2: where a tile's cycles go in the re-sequenced kernels (stamps), and its floors (no memory / no GELU / no MFMA)
R=$GRAFT_REPO_ROOT; cd $R
NAMP_LIB_PATH=$R/tools/_variants/stamps.so timeout 300 python tools/p32_stamps.py 2>&1 | grep -v Warning
for v in p_nomfma p_nomem p_nomem_nogelu p_skel; do NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 300 python tools/cfg3_ab.py --masks 3 --reps 1 2>&1 | grep mask; done
