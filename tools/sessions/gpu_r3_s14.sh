# Round 3, session 14: the split-bf16 persistent edge kernel with 8 waves per workgroup (256 VGPRs) and fragments requested ahead: cfg3 x3, cfg4
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for v in base w8 w8a; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg3 --precision x3 --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg3-x3', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items()})"
done
done
for v in base w8a; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg4', d['ms_per_step'], d['value'])"
done
