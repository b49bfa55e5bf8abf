# Round 4, session 7: branch-free stationary backward — kernel time, gradient tests, cfg5 A/B
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1; do python tools/dw_time.py --prec 2 --mode $m; done
python tools/dw_time.py --prec 2 --mode 0 --old
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3
run() {
  env $1 timeout 600 python bench.py --workload cfg5 --precision $2 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d.get('ms_per_step'), d.get('hip_kernel_share'), d.get('whole_step'))"
}
for rep in 1 2; do run NAMP_TRAIN_DW=0 bf16; run NAMP_TRAIN_DW=1 bf16; done
