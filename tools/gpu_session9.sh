export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'x3', d['x3']['value'], d['x3']['ms_per_step'], d['roofline']['frac'], d.get('parity_vs_cpu') or d.get('parity'))
print({k:v['avg_ms'] for k,v in d['per_kernel'].items()})
for s in d['secondary']:
    print(s['config']['workload'][:50], s['value'], s['ms_per_step'], s.get('hip_kernel_share'))
    if 'mixed_precision_bf16' in s: print('   mixed', s['mixed_precision_bf16']['ms_per_step'])
PY
timeout 900 python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python -c "
import json
d=json.loads(open('$O/bench_cfg4.json').read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'])"
