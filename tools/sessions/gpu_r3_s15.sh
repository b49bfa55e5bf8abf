# Round 3, session 15: the sampler's tile GEMMs with fragments requested ahead (default) vs the compiler's schedule (NAMP_SAMPLE_AHEAD=0)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for v in base noahead; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg1', d['ms_per_step'], d['value'])"
done
done
unset NAMP_LIB_PATH
timeout 300 python tools/sample_time.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "sample or symmetric or pair_bias or sampler" 2>&1 | tail -3
