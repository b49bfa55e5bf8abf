# round 5, session 1: where the 20-step cfg2 region's fixed cost goes (HEAD vs the round-3 tree, same box), then same-box baselines
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s1
for i in 1 2; do
  for prec in fp32 x3; do
    (cd $R && timeout 300 python tools/region_probe.py $prec head 2>/dev/null | tail -1) >> gpurun_out/r5s1/region.jsonl
    (cd $R/tools/_variants/r3tree && timeout 300 python $R/tools/region_probe.py $prec r3 2>/dev/null | tail -1) >> gpurun_out/r5s1/region.jsonl
  done
done
cat gpurun_out/r5s1/region.jsonl
# driver-style lines, alternating trees
for i in 1 2 3; do
  (cd $R && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('head', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'))")
  (cd $R/tools/_variants/r3tree && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r3', d['ms_per_step'], (d.get('x3') or {}).get('ms_per_step'))")
done
# same-box baselines of the other workloads (HEAD)
timeout 600 python tools/train_time.py --steps 6 --precision bf16 2>&1 | tail -25 > gpurun_out/r5s1/train_bf16.txt
timeout 600 python tools/train_time.py --steps 6 2>&1 | tail -25 > gpurun_out/r5s1/train_x3.txt
tail -8 gpurun_out/r5s1/train_bf16.txt; tail -8 gpurun_out/r5s1/train_x3.txt
timeout 300 python tools/sample_time.py 2>&1 | tail -12
