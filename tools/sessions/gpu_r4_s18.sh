# full GPU suite + default bench with the 4x4x1-MFMA residue tail
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r4_s18; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 1500 $O/bench_default.json; echo
