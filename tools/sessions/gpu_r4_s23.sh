cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04f
S=$(date +%s); timeout 900 python bench.py > gpurun_out/r04f/bench_default.json 2> gpurun_out/r04f/bench_default.err; echo "wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f/bench_default.json"))
print(d["ms_per_step"], d["x3"]["ms_per_step"])
for s in d["secondary"]:
    print(s.get("workload"), s.get("ms_per_step"), s.get("steps"), (s.get("mixed_precision_bf16") or s.get("mixed") or {}))
PY
