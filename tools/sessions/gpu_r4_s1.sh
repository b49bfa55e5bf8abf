# Round 4, session 1: bf16 mode — plain-bf16 residue GEMMs (default) vs split (NAMP_BF16S_RESIDUE_X3=1), GELU degree 4 (base) vs 6 / 2
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
run() {
  timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-gather --no-secondary --no-pmc --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in d['per_kernel'].items()})"
}
for rep in 1 2; do
  unset NAMP_LIB_PATH; unset NAMP_BF16S_RESIDUE_X3
  run base_deg4_x1
  NAMP_BF16S_RESIDUE_X3=1 run deg4_x3
  export NAMP_LIB_PATH=$R/tools/_variants/gelu6.so
  run deg6_x1
  NAMP_BF16S_RESIDUE_X3=1 run deg6_x3_round3
  export NAMP_LIB_PATH=$R/tools/_variants/gelu2.so
  run deg2_x1
  unset NAMP_LIB_PATH
done
echo "--- accuracy: deg4 x1 (shipped)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "cfg3 or bf16" 2>&1 | grep -E "max\|dlogp\||bf16s32|passed|failed|Error" | tail -20
echo "--- accuracy: deg2 x1"
NAMP_LIB_PATH=$R/tools/_variants/gelu2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "cfg3_sized or bf16_throughput" 2>&1 | grep -E "max\|dlogp\||passed|failed" | tail -8
echo "--- accuracy: deg4 x3"
NAMP_BF16S_RESIDUE_X3=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "cfg3_sized" 2>&1 | grep -E "max\|dlogp\||passed|failed" | tail -4
