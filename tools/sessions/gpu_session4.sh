set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
tail -6 $O/pytest_train.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or cfg3" > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
tail -4 $O/pytest_bf16.log
for p in x3 bf16; do
timeout 600 python bench.py --workload cfg5 --precision $p --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_$p.json 2> $O/bench_cfg5_$p.err
done
timeout 600 python bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-gather > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02d/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d.get('hip_kernel_share'))
        if 'other_kernels_ms_per_step' in d: print({k[:40]:v for k,v in list(d['other_kernels_ms_per_step'].items())[:6]})
        if 'per_kernel' in d: print({k:v['ms_per_step'] for k,v in d['per_kernel'].items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
