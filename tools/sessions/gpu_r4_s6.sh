# Round 4, session 6: issue-side counters of the stationary backward (where do a wave's cycles go?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_s6
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $O/p1 -o k -- python $R/tools/dw_time.py --prec 2 --mode 0 --reps 3 > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $O/p2 -o k -- python $R/tools/dw_time.py --prec 2 --mode 0 --reps 3 > $O/p2.log 2>&1
cd $R
for p in p1 p2; do python tools/rocpd_pmc.py $(ls $O/$p/*/*.db $O/$p/*.db 2>/dev/null | head -1) edge_bwd_dw; tail -2 $O/$p.log; done
rm -rf $O/p1 $O/p2
