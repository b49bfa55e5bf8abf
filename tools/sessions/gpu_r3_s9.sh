# Round 3, session 9: same-box A/B of the cfg3 step: HEAD tree (_old_tree, 16x16x32 bf16-storage kernels) vs working tree (32x32x16)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s9
mkdir -p $O
for rep in 1 2; do
for t in old new; do
  if [ $t = old ]; then D=$R/_old_tree; else D=$R; fi
  cd $D
  timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose > $O/bench_${t}_$rep.log 2>&1
  python - $O/bench_${t}_$rep.log $t <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})
PY
done
done
