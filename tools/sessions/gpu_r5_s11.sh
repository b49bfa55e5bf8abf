R=$GRAFT_REPO_ROOT; cd $R
for g in 0 1 2 4 8; do
  if [ $g = 0 ]; then unset NAMP_WALK_GRID; else export NAMP_WALK_GRID=$g; fi
  echo "== grid $g"; timeout 300 python tools/sample_profile.py 2>&1 | grep "dec_sample_kernel\|host enqueue" | cut -c1-130
done
