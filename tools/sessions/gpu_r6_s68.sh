R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
rm -rf /tmp/p1; (cd /tmp; timeout 900 rocprofv3 --kernel-trace -d /tmp/p1 -o c1 -- python $R/bench.py --workload cfg1 --steps 6 --warmup 3 --no-cpu-baseline > /tmp/p1.log 2>&1)
python tools/rocpd_timeline.py $(db /tmp/p1) 64
