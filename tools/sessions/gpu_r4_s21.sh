R=$GRAFT_REPO_ROOT; cd $R
for st in 20 50 200 1000 50 20; do python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $st', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'), d['roofline'].get('clock_mhz'))"; done
