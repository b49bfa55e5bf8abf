R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  NAMP_LIB_PATH=$R/tools/_variants/l2pf.so run prev
  run new
  NAMP_LIB_PATH=$R/tools/_variants/ln1w0.so run ln1w0
  NAMP_LIB_PATH=$R/tools/_variants/latemeta.so run latemeta
done
