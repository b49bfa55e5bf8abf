R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16p_equals or bf16_storage_message or bf16_throughput or cfg3_sized" 2>&1 | tail -3
timeout 600 python tools/cfg3_ab.py --masks 0,3,7 --reps 2 2>&1 | grep mask
