cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-gather --no-secondary --no-pmc 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.readline()); print('cfg2 fp32', o['ms_per_step'], 'x3', o['x3']['ms_per_step'])"
  HIP_FORCE_DEV_KERNARG=$v python bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.readline()); print('cfg1', o['ms_per_step'])"
  HIP_FORCE_DEV_KERNARG=$v NAMP_LIB_PATH=$GRAFT_REPO_ROOT/tools/_variants/stamps.so python tools/sample_stamps.py 2>/dev/null | grep -A14 "batch_size=1 launch"
done
