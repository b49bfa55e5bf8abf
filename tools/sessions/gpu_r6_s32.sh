R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
NAMP_LIB_PATH=$R/tools/_variants/feat_stamps.so timeout 600 python tools/feat_stamps.py 2>&1 | grep -v amdgpu.ids
for m in 11 43; do
  rm -rf /tmp/fp_$m; FEAT_MASKS=$m timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/fp_$m -o fp -- python tools/feat_parts_ab.py > /dev/null 2>&1
  f=$(find /tmp/fp_$m -name "*kernel_stats.csv" | head -1)
  echo "== mask $m kernel stats"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
done
for m in 11 43; do
NAMP_BF16P=$m timeout 600 python tools/score_ab.py 2>&1 | grep "cfg2 side_stream=True" | sed "s/^/mask $m /"
NAMP_BF16P=$m timeout 600 python bench.py --workload cfg1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $m cfg1', d['ms_per_step'], d.get('latency_ms'))"
done
