export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof4 -o cfg4 -- python $R/bench.py --workload cfg4 --no-cpu-baseline > $O/prof4.log 2>&1
cd $R
db=$(ls $O/prof4/*/*.db $O/prof4/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $db > $O/cfg4_kernel_stats.md
rm -rf $O/prof4
head -30 $O/cfg4_kernel_stats.md
tail -2 $O/prof4.log | cut -c1-300
