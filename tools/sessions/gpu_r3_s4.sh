# Round 3, session 4: VALU rate probe, sampler attribution (ablation builds of the r2 kernel vs the r3 kernel), cfg1 kernel trace, tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s4
mkdir -p $O
cd $R
tools/_variants/valu_rate_probe > $O/valu_rate_probe.md 2>&1
for v in default s_notail s_nogemm s_nodma s_notail_nogemm_nodma; do
  lib=$R/tools/_variants/$v.so; [ $v = default ] && lib=$R/na_mpnn_amd/lib/libnamp_hip.so
  echo "== $v" >> $O/sample_time.txt
  NAMP_LIB_PATH=$lib timeout 300 python tools/sample_time.py >> $O/sample_time.txt 2>&1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o cfg1 -- python $R/bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline > $O/prof1.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/prof1/*/*.db $O/prof1/*.db 2>/dev/null | head -1) > $O/cfg1_kernel_stats.md 2>&1
rm -rf $O/prof1
timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg1.json 2>$O/bench_cfg1.err
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -x -q -k "sample or extreme or mixed_precision" -s 2>&1 | tail -12
cat $O/valu_rate_probe.md; cat $O/sample_time.txt; head -30 $O/cfg1_kernel_stats.md; cat $O/bench_cfg1.json
