R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r5s13
python bench.py --steps 20 --warmup 5 > gpurun_out/r5s13/bench_default.json 2> gpurun_out/r5s13/bench_default.err
tail -c 6000 gpurun_out/r5s13/bench_default.json
