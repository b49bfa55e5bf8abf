set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01l
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for w in cfg1 cfg3 cfg4 cfg5; do timeout 900 python bench.py --workload $w --no-cpu-baseline >> $O/bench_all.json 2>> $O/bench_all.err; done
timeout 300 python bench.py --precision fp32 --no-cpu-baseline >> $O/bench_all.json 2>> $O/bench_all.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gather > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gather > $O/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof5 -o train -- python $R/bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof5.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) > $O/kernel_stats.md
python tools/rocpd_summary.py $(ls $O/prof5/*/*.db $O/prof5/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.md
python tools/rocpd_pmc.py $(ls $O/pmc_f/*/*.db $O/pmc_f/*.db 2>/dev/null | head -1) edge_mlp node_ > $O/pmc_fetch.txt
python tools/rocpd_pmc.py $(ls $O/pmc_w/*/*.db $O/pmc_w/*.db 2>/dev/null | head -1) edge_mlp node_ > $O/pmc_write.txt
rm -rf $O/prof $O/prof5 $O/pmc_f $O/pmc_w
tail -3 $O/pytest.log; cat $O/bench_cfg2.json
