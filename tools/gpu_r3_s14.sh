# Round 3, session 14: hi-fragment prefetch in the split-bf16 chain GEMM (NAMP_X3_PREFETCH = sched_barrier mask): cfg3 x3, cfg2 x3, cfg4
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for v in base x3p2 x3p10; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  a=$(timeout 600 python bench.py --workload cfg3 --precision x3 --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  b=$(timeout 600 python bench.py --workload cfg2 --precision x3 --steps 50 --warmup 5 --no-cpu-baseline --no-gather --no-secondary --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "$v cfg3-x3 $a ms  cfg2-x3 $b ms"
done
done
