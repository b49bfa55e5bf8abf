# Round 3, session 8: bf16-storage path on 32x32x16 MFMA — parity (cfg3 tests), kernel check, cfg3 bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg3 or bf16 or storage" -s > $O/parity.log 2>&1; echo "parity rc=$?"; tail -15 $O/parity.log
timeout 300 python tools/bf16s32_check.py > $O/check.log 2>&1; tail -8 $O/check.log
timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose > $O/bench_cfg3.log 2>&1; tail -c 3000 $O/bench_cfg3.log
