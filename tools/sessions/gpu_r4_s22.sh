R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "edge_update_backward" 2>&1 | tail -15
for e in 0 1; do NAMP_TRAIN_DW_EDGE=$e python bench.py --workload cfg5 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dw_edge=$e bf16', d['ms_per_step'])"; done
