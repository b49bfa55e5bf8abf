R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/feat_parts_ab.py 2>&1 | grep cfg2
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r6_s35_bench.json; head -c 1500 gpurun_out/r6_s35_bench.json
