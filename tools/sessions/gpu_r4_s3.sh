# Round 4, session 3: weight-stationary bf16 backward (dw16) vs ring form vs round-3 form; per-kernel stats of a cfg5 step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r4_s3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "edge_mlp_backward" 2>&1 | grep -E "on-chip|passed|failed|Error|error" | tail -20
run() {
  env $1 timeout 600 python bench.py --workload cfg5 --precision $2 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d.get('ms_per_step'), d.get('whole_step',{}).get('final_loss'))"
}
for rep in 1 2; do
  run NAMP_TRAIN_DW=0 bf16; run NAMP_TRAIN_DW=1 bf16; run NAMP_DW16_RING=1 bf16; run NAMP_TRAIN_DW=0 x3; run NAMP_TRAIN_DW=1 x3
done
cd /tmp
for p in bf16 x3; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$p -o train -- python $R/bench.py --workload cfg5 --precision $p --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/prof_$p.log 2>&1
  cd $R
  python tools/rocpd_summary.py $(ls $O/prof_$p/*/*.db $O/prof_$p/*.db 2>/dev/null | head -1) > $O/train_${p}_kernel_stats.md
  rm -rf $O/prof_$p
  head -16 $O/train_${p}_kernel_stats.md
  cd /tmp
done
