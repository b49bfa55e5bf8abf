R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  run base
  NAMP_LIB_PATH=$R/tools/_variants/nogelu.so run nogelu
done
