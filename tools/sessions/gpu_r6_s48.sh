R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>gpurun_out/r6_s48_bench.err | tail -1 > gpurun_out/r6_s48_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_s48_bench.json').read())
print({k:d[k] for k in ('value','ms_per_step','steps')}); print(d['features']); print([ (s['workload'], s['ms_per_step'], s.get('roofline',{}).get('frac_at_measured_clock')) for s in d['secondary']])
print(d['cpu_baseline'].get('gpu_full_forward_ms'))
PY
