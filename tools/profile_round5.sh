# Round profile set (round 4 on): default bench line, rocprofv3 kernel-trace stats (cfg2, cfg3, cfg5 x3 / bf16, featuriser), PMC traffic (cfg2 + cfg5),
# issue counters of the cfg3 launches.      gpurun --timeout 3000 -- 'bash tools/profile_round5.sh <tag, e.g. r05f> <git commit>'
export TMPDIR=/tmp
TAG=${1:-r04}
COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-pmc"
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- $B --steps 50 --warmup 5 > $O/prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o k -- $B --steps 10 --warmup 2 > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o k -- $B --steps 10 --warmup 2 > $O/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pmc > $O/prof3.log 2>&1
for p in x3 bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof5_$p -o train -- python $R/bench.py --workload cfg5 --precision $p --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/prof5_$p.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc5f_$p -o k -- python $R/bench.py --workload cfg5 --precision $p --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/pmc5f_$p.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc5w_$p -o k -- python $R/bench.py --workload cfg5 --precision $p --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/pmc5w_$p.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/proff -o feat -- python $R/tools/feat_time.py > $O/proff.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof4 -o cfg4 -- python $R/bench.py --workload cfg4 --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/bench_cfg4.json 2> $O/prof4.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_sq_bf16 -o k -- python $R/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-pmc > $O/pmc_sq_bf16.log 2>&1
cd $R
python tools/rocpd_summary.py $(db $O/prof) > $O/bench_cfg2_kernel_stats.md
python tools/rocpd_summary.py $(db $O/prof3) > $O/bench_cfg3_kernel_stats.md
python tools/rocpd_summary.py $(db $O/prof5_x3) > $O/train_cfg5_x3_kernel_stats.md
python tools/rocpd_summary.py $(db $O/prof5_bf16) > $O/train_cfg5_bf16_kernel_stats.md
python tools/rocpd_summary.py $(db $O/proff) > $O/featurizer_kernel_stats.md
python tools/rocpd_summary.py $(db $O/prof4) > $O/bench_cfg4_kernel_stats.md
python tools/make_pmc_traffic.py $(db $O/pmc_f) $(db $O/pmc_w) $COMMIT $TAG $(db $O/pmc5f_x3) $(db $O/pmc5w_x3) x3 $(db $O/pmc5f_bf16) $(db $O/pmc5w_bf16) bf16 > $O/pmc_traffic.json
python tools/rocpd_pmc.py $(db $O/pmc_sq_bf16) edge_mlp node_update node_linear > $O/pmc_issue_counters_cfg3.txt
rm -rf $O/prof4 $O/prof $O/prof3 $O/prof5_x3 $O/prof5_bf16 $O/proff $O/pmc_f $O/pmc_w $O/pmc5f_x3 $O/pmc5w_x3 $O/pmc5f_bf16 $O/pmc5w_bf16 $O/pmc_sq_bf16
head -c 600 $O/bench_default.json; echo; head -12 $O/train_cfg5_bf16_kernel_stats.md; python -c "
import json; d=json.load(open('$O/pmc_traffic.json'))
for k in ('cfg5_x3','cfg5_bf16'):
    print(k); [print('  ',n,v) for n,v in d.get(k,{}).items() if 'algorithmic_bytes' in v]"
python tools/spills.py > $O/register_report.txt
