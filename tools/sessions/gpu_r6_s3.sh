# round 6, session 3: address-pattern probe of the vector-memory path; the re-sequenced kernels with two-stage metadata and no LDS-DMA in the loop
R=$GRAFT_REPO_ROOT; cd $R
timeout 120 tools/_variants/ta_probe
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16p_equals or bf16_storage_message" 2>&1 | tail -3
timeout 600 python tools/cfg3_ab.py --masks 0,3 --reps 2 2>&1 | grep mask
NAMP_LIB_PATH=$R/tools/_variants/stamps.so timeout 300 python tools/p32_stamps.py 2>&1 | grep -v Warning
