R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "node_update_w or bf16_throughput or cfg3_sized or bf16p_equals" 2>&1 | grep "vs exact\|bf16\|passed\|failed\|Error\|assert"
