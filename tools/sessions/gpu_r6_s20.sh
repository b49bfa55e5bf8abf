R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "node_update_w or cfg3_sized or encoder_decoder_goldens or bf16_throughput" 2>&1 | grep "split-bf16 residue\|passed\|failed\|Error\|assert" | head
for i in 1 2; do
NAMP_BF16P=3 timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 old residue update', d['ms_per_step'])"
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 node_update_w<x3>', d['ms_per_step'])"
done
