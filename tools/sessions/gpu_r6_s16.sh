R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s16_prof4 -o cfg4 -- python $R/bench.py --workload cfg4 --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $(ls gpurun_out/s16_prof4/*/*.db gpurun_out/s16_prof4/*.db 2>/dev/null | head -1) | head -40 | cut -c1-150; rm -rf gpurun_out/s16_prof4
