# Round 4, session 2: first run of the persistent message backward with on-chip weight gradients
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "edge_mlp_backward" 2>&1 | tail -25
echo "--- cfg5 A/B (ms per step: x3, then bf16 mixed precision)"
run() {
  NAMP_TRAIN_DW=$1 timeout 600 python bench.py --workload cfg5 --precision $2 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dw=$1 $2', d.get('ms_per_step'), {k:d[k] for k in ('peak_mem_gib','hip_kernel_share') if k in d}, d.get('whole_step',{}).get('final_loss'))"
}
for rep in 1 2; do
  run 0 x3; run 1 x3; run 0 bf16; run 1 bf16
done
