# Round 3, session 1: co-execution probe for the 32x32x16 / fp32 MFMA shapes + A/B of SLP packing and the bf16-mode GELU form.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r3_s1.sh'
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s1
mkdir -p $O
tools/_variants/coexec_probe3 > $O/probe3_noslp.md 2>&1
tools/_variants/coexec_probe3_slp > $O/probe3_slp.md 2>&1
V=$GRAFT_REPO_ROOT/tools/_variants
run() {  # name lib workload precision
  NAMP_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --precision $4 --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-secondary 2>>$O/err.log |
    python -c "import json,sys; o=json.loads(sys.stdin.readline()); pk=o.get('per_kernel',{}); print('$1 $3 $4', o['ms_per_step'], ' '.join(f'{k}={v[\"avg_ms\"]*1e3:.1f}' for k,v in pk.items()))" >> $O/ab.txt
}
D=$GRAFT_REPO_ROOT/na_mpnn_amd/lib/libnamp_hip.so
for rep in 1 2; do
  for v in default noslp gelu16s gelu16s_noslp; do
    lib=$V/$v.so; [ $v = default ] && lib=$D
    run $v $lib cfg3 bf16
  done
  for v in default noslp; do
    lib=$V/$v.so; [ $v = default ] && lib=$D
    run $v $lib cfg3 x3
    run $v $lib cfg2 x3
    run $v $lib cfg2 fp32
  done
done
for v in default noslp; do
  lib=$V/$v.so; [ $v = default ] && lib=$D
  NAMP_LIB_PATH=$lib timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>>$O/err.log | python -c "import json,sys; o=json.loads(sys.stdin.readline()); print('$v cfg5', o['ms_per_step'], o['hip_kernel_share'])" >> $O/ab.txt
done
cat $O/probe3_noslp.md; cat $O/ab.txt
