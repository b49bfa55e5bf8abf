R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_s11_bench.json 2> gpurun_out/r6_s11_bench.err; tail -c 6000 gpurun_out/r6_s11_bench.json
