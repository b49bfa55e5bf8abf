export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or cfg3" 2>&1 | tail -4
timeout 600 python bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-gather > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02g/bench_cfg3.json')); print(d['ms_per_step'], d['value']); print({k:v['ms_per_step'] for k,v in d['per_kernel'].items()})
PY
