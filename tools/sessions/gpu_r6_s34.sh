R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
NAMP_LIB_PATH=$R/tools/_variants/feat_stamps.so timeout 600 python tools/feat_stamps.py 2>&1 | grep -v amdgpu.ids | grep "==\|body\|set-up"
timeout 600 python tools/feat_parts_ab.py 2>&1 | tail -16
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
for m in 43 171; do
  rm -rf /tmp/fp_$m; (cd /tmp; FEAT_MASKS=$m timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/fp_$m -o fp -- python $R/tools/feat_parts_ab.py > /tmp/fp_$m.log 2>&1)
  echo "== mask $m kernel stats"; python tools/rocpd_summary.py $(db /tmp/fp_$m) | head -7
done
