export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM -d $O/pmc_wait -o k -- python $R/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-gather > $O/pmc_wait.log 2>&1
cd $R
python tools/rocpd_pmc.py $(ls $O/pmc_wait/*/*.db $O/pmc_wait/*.db 2>/dev/null | head -1) edge_mlp node_update > $O/pmc_wait_bf16.txt
rm -rf $O/pmc_wait
cat $O/pmc_wait_bf16.txt; tail -3 $O/pmc_wait.log
