export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err ) 2>&1 | grep real
head -c 600 gpurun_out/final/bench.json
