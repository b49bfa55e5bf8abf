# featuriser weight gradient, 4 blocks per wave: parity + per-kernel time in the training step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4_s19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "feature_weight" 2>&1 | tail -5
cd /tmp
for p in bf16 x3; do
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof5_$p -o train -- python $R/bench.py --workload cfg5 --precision $p --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/prof5_$p.log 2>&1
python $R/tools/rocpd_summary.py $(ls $O/prof5_$p/*/*.db $O/prof5_$p/*.db 2>/dev/null | head -1) > $O/train_cfg5_${p}_kernel_stats.md
rm -rf $O/prof5_$p
grep -n "feat_wgrad\|edge_features" $O/train_cfg5_${p}_kernel_stats.md
done
cd $R
for p in bf16 x3; do python bench.py --workload cfg5 --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', d['ms_per_step'])"; done
