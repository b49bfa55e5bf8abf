# same-box A/B of cfg3: HEAD's library (tools/_variants/head.so) against the working tree's
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --workload cfg3 --steps 30 --warmup 5 --no-cpu-baseline --no-gather --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3 4; do
  NAMP_LIB_PATH=$R/tools/_variants/head.so run head
  run new
done
