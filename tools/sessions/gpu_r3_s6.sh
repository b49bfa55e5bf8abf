# Round 3, session 6: what stalls the sampler's workgroup?  SQ / SQC counters of dec_sample_kernel (cfg1) and of the cfg2 fused launches
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s6
mkdir -p $O
cd /tmp
pass() {  # tag workload counters...
  tag=$1; wl=$2; shift; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o k -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-secondary --no-pmc > $O/pmc_$tag.log 2>&1
  db=$(ls $O/pmc_$tag/*/*.db $O/pmc_$tag/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_pmc.py $db dec_sample edge_mlp_kernel node_linear > $O/pmc_$tag.txt 2>&1
  rm -rf $O/pmc_$tag
}
for wl in cfg1 cfg2; do
  pass ${wl}_wait $wl SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU
  pass ${wl}_ifetch $wl SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA
  pass ${wl}_icache $wl SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL
done
cat $O/pmc_cfg1_wait.txt | head -12; cat $O/pmc_cfg1_ifetch.txt | head -12; cat $O/pmc_cfg1_icache.txt | head -12
