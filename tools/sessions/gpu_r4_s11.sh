# Round 4, session 11: where is the device idle during a cfg5 training step? (kernel-trace gaps)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_s11
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/p -o t -- python $R/tools/train_time.py --precision bf16 --steps 10 > $O/p.log 2>&1
cd $R
python tools/rocpd_gaps.py $(ls $O/p/*/*.db $O/p/*.db 2>/dev/null | head -1) --top 25 --min-gap-us 20 --last-ms 250
tail -3 $O/p.log
rm -rf $O/p
