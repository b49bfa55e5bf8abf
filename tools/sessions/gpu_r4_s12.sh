# bf16-storage message kernels without LDS-DMA in the step loop: parity + cfg3 per-kernel times
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s12; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16" 2>&1 | tail -5
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pmc > $O/prof3.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/prof3/*/*.db $O/prof3/*.db 2>/dev/null | head -1) > $O/cfg3_kernel_stats.md
rm -rf $O/prof3
head -16 $O/cfg3_kernel_stats.md
for i in 1 2 3; do python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('accuracy'))"; done
