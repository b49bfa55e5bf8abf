# round 5, session 2: the two-launch edge-update backward — parity, then timing against the round-3 form
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s2
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "on_chip or fp64 or edge_update or mixed_precision" 2>&1 | tail -25
for v in 1 0; do
  echo "== NAMP_TRAIN_DW_EDGE=$v"
  NAMP_TRAIN_DW_EDGE=$v timeout 600 python tools/train_time.py --steps 6 --precision bf16 --profile 2>&1 | tail -32 > gpurun_out/r5s2/train_bf16_edge$v.txt
  grep -E "ms/step|edge_update|edge_chain|wgrad|scatter" gpurun_out/r5s2/train_bf16_edge$v.txt | head -20
done
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
