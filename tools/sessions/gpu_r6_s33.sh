R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
for m in 11 43; do
  rm -rf /tmp/fp_$m; (cd /tmp; FEAT_MASKS=$m timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/fp_$m -o fp -- python $R/tools/feat_parts_ab.py > /tmp/fp_$m.log 2>&1)
  echo "== mask $m kernel stats"; python tools/rocpd_summary.py $(db /tmp/fp_$m) | head -12
done
