# round 6, session 1: the re-sequenced bf16-storage edge launches (namp_bf16p.h): bit-equality with the round-3 kernels, A/B inside the cfg3 step,
# ablations of both forms
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16p_equals or bf16_storage_message" 2>&1 | tail -5
timeout 600 python tools/cfg3_ab.py --masks 0,3,7 --reps 2 2>&1 | grep -v Warning
for v in nofence dist2 dist5; do NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 300 python tools/cfg3_ab.py --masks 3 --reps 1 2>&1 | grep mask; done
for v in nogelu nomfma noldsw; do NAMP_LIB_PATH=$R/tools/_variants/$v.so timeout 300 python tools/cfg3_ab.py --masks 0,3 --reps 1 2>&1 | grep mask; done
