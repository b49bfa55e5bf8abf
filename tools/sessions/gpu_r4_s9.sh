# Round 4, session 9: full GPU suite + the default bench line (new: features, cfg1s, clock, cfg4_strong at N = 2)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4_s9
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4_s9/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_s9/pytest.log
tail -15 gpurun_out/r4_s9/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_s9/bench_default.json 2> gpurun_out/r4_s9/bench_default.err
cat gpurun_out/r4_s9/bench_default.json | head -c 7000; tail -5 gpurun_out/r4_s9/bench_default.err
