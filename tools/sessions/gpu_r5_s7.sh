R=$GRAFT_REPO_ROOT; cd $R
for v in head nogelu notail nogemm; do
  if [ $v = head ]; then unset NAMP_LIB_PATH; else export NAMP_LIB_PATH=$R/tools/_variants/$v.so; fi
  echo "== $v"; timeout 300 python tools/sample_profile.py 2>&1 | grep "dec_sample_kernel\|host enqueue" | cut -c1-150
done
