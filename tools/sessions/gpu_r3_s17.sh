# Round 3, session 17: LayerNorm statistics / K-sums as packed pairs in the bf16-storage launches: previous build (prev) vs this one (base)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for v in prev base; do
  if [ $v = base ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['per_kernel'].items() if 'e' in k and k!='node_update'})"
done
done
unset NAMP_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg3 or bf16" 2>&1 | tail -3
