# Round 3, session 2: wait-state / issue breakdown of the large-batch edge launches (cfg3 bf16) and the cfg5 backward launch.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s2
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
B="python $R/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-pmc"
pass() {  # tag, counters...
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o k -- $B > $O/pmc_$tag.log 2>&1
  db=$(ls $O/pmc_$tag/*/*.db $O/pmc_$tag/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_pmc.py $db edge_mlp node_update > $O/pmc_$tag.txt 2>&1
  rm -rf $O/pmc_$tag
}
pass wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES
pass act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
pass lds SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
B="python $R/bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline"
pass5() {
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o k -- $B > $O/pmc_$tag.log 2>&1
  db=$(ls $O/pmc_$tag/*/*.db $O/pmc_$tag/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_pmc.py $db edge_chain wgrad scatter tail_train > $O/pmc_$tag.txt 2>&1
  rm -rf $O/pmc_$tag
}
pass5 t_wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
head -50 $O/pmc_wait.txt; tail -3 $O/pmc_wait.log
cd $R
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
wc -c $O/bench_default.json; head -c 6200 $O/bench_default.json; tail -5 $O/bench_default.err
timeout 2400 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -15
