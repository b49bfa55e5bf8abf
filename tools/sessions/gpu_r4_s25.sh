R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'))"; }
for i in 1 2 3; do
  NAMP_LIB_PATH=$R/tools/_variants/fullsync.so run fullsync
  run new
done
for v in fullsync new; do echo == $v; if [ $v = fullsync ]; then export NAMP_LIB_PATH=$R/tools/_variants/fullsync.so; else unset NAMP_LIB_PATH; fi; timeout 300 python tools/sample_time.py 2>&1 | grep "sample()" | cut -c1-60; done
unset NAMP_LIB_PATH
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
