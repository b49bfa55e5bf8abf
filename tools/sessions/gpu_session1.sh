set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
tools/_variants/coexec_probe2 > $O/coexec_probe2.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/pytest.log; cat $O/coexec_probe2.txt; cat $O/bench_default.json | head -c 6000
