# Round 3, session 16: the training backward launch with fragments requested ahead (default) vs the compiler's schedule
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for v in base bwd_noahead; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg5', d['ms_per_step'], d['value'])"
done
done
unset NAMP_LIB_PATH
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3
