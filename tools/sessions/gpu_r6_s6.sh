R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "node_update_w or bf16p_equals or bf16_throughput or cfg3_sized" 2>&1 | grep -v "^$" | tail -12
timeout 600 python tools/cfg3_ab.py --masks 3,11 --reps 2 2>&1 | grep mask
