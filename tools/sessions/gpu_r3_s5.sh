export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s5
mkdir -p $O
cd $R
timeout 300 python tools/sample_time.py > $O/sample_time.txt 2>&1
timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg1.json 2>$O/bench_cfg1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o cfg1 -- python $R/bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline > $O/prof1.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/prof1/*/*.db $O/prof1/*.db 2>/dev/null | head -1) > $O/cfg1_kernel_stats.md 2>&1
rm -rf $O/prof1
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -x -q -k "sample or extreme or cli or level" -s 2>&1 | tail -12
cat $O/sample_time.txt; head -8 $O/cfg1_kernel_stats.md; cut -c1-400 $O/bench_cfg1.json
