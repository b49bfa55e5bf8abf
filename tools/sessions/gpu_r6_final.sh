# end-of-round validation: full GPU suite, smoke, default bench, profile set
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round5.sh r06h 604f1c5 > gpurun_out/r06h.log 2>&1; tail -25 gpurun_out/r06h.log
