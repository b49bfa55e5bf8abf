R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "bf16p_equals or node_update_w or bf16_throughput or cfg3_sized" 2>&1 | grep "one-pass\|vs exact\|cfg3-sized bf16\|passed\|failed\|Error"
timeout 600 python tools/cfg3_ab.py --masks 27,11 --reps 3 2>&1 | grep mask
