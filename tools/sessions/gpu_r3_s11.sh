# Round 3, session 11: phase skew between the two waves of a SIMD in the bf16-storage launches (NAMP_BF16S_SKEW x 1,024 cycles)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for v in base skew2 skew3 skew5 skew8; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items() if 'e' in k and k!='node_update'})"
done
done
