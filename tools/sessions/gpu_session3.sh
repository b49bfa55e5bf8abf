set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
tail -8 $O/pytest_train.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "cli or noN or pred_na" > $O/pytest_cli.log 2>&1; echo "rc=$?" >> $O/pytest_cli.log
tail -8 $O/pytest_cli.log
for p in x3 bf16 fp32; do
timeout 600 python bench.py --workload cfg5 --precision $p --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_$p.json 2> $O/bench_cfg5_$p.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02c/bench_cfg5_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d['hip_kernel_share'], d['whole_step']); print({k:v for k,v in list(d['per_kernel_ms_per_step'].items())[:12]})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
