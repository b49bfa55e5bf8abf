R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['x3']['ms_per_step'], d['roofline']['avg_launch_ms'], d['x3']['avg_launch_ms'])"; }
for i in 1 2 3; do
  NAMP_LIB_PATH=$R/tools/_variants/noxcd.so run noxcd
  run xcd
done
