# Round profile: tests, the default bench line, rocprofv3 kernel-trace stats and PMC passes.  Run on the GPU box:
#   gpurun --timeout 3000 -- 'bash tools/profile_round.sh <round tag, e.g. r02e> <git commit>'
set -x
export TMPDIR=/tmp
TAG=${1:-r02}
COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
for w in cfg4; do timeout 900 python bench.py --workload $w --no-cpu-baseline --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; done
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-pmc"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- $B --steps 50 --warmup 5 > $O/prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o k -- $B --steps 10 --warmup 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o k -- $B --steps 10 --warmup 2 > $O/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof5 -o train -- python $R/bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/prof5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pmc > $O/prof3.log 2>&1
# issue-side counters of the large-batch persistent kernels (bf16 and x3 at B = 64)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/pmc_sq_bf16 -o k -- python $R/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-pmc > $O/pmc_sq_bf16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/pmc_sq_x3 -o k -- python $R/bench.py --workload cfg3 --precision x3 --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-pmc > $O/pmc_sq_x3.log 2>&1
cd $R
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
python tools/rocpd_summary.py $(db $O/prof) > $O/kernel_stats.md
python tools/rocpd_summary.py $(db $O/prof5) > $O/train_kernel_stats.md
python tools/rocpd_summary.py $(db $O/prof3) > $O/cfg3_kernel_stats.md
python tools/make_pmc_traffic.py $(db $O/pmc_f) $(db $O/pmc_w) $COMMIT $TAG > $O/pmc_traffic.json
python tools/rocpd_pmc.py $(db $O/pmc_sq_bf16) edge_mlp node_update > $O/pmc_sq_bf16.txt
python tools/rocpd_pmc.py $(db $O/pmc_sq_x3) edge_mlp node_update > $O/pmc_sq_x3.txt
rm -rf $O/prof $O/prof5 $O/prof3 $O/pmc_f $O/pmc_w $O/pmc_sq_bf16 $O/pmc_sq_x3
tail -3 $O/pytest.log; head -c 1500 $O/bench_default.json; cat $O/pmc_sq_bf16.txt | head -40
