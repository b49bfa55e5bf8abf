R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q 2>&1 | tail -4
timeout 600 python tools/cfg3_ab.py --masks 3,11 --reps 2 2>&1 | grep mask
for m in 3 11; do NAMP_BF16P=$m timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 mask $m', d['ms_per_step'])"; done
for m in 3 11; do NAMP_BF16P=$m timeout 600 python bench.py --workload cfg5 --precision bf16 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 bf16 mask $m', d['ms_per_step'])"; done
