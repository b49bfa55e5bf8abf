R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
rm -rf /tmp/ps; (cd /tmp; timeout 600 rocprofv3 --kernel-trace -d /tmp/ps -o sc -- python $R/tools/score_once.py > /tmp/ps.log 2>&1)
python tools/rocpd_timeline.py $(db /tmp/ps) 48
