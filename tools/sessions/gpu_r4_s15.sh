# 4x4x1-MFMA residue tail: parity (cfg2 goldens, sampler forms) + cfg2 / cfg1 timing against HEAD's library on the same box
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sample.py -x -q 2>&1 | tail -8
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  NAMP_LIB_PATH=$R/tools/_variants/head.so run head
  run new
done
for w in cfg1 cfg1s; do
  python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w new', d['ms_per_step'])"
  NAMP_LIB_PATH=$R/tools/_variants/head.so python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w head', d['ms_per_step'])"
done
