export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 3300 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/full/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/full/pytest.log 2>&1
tail -25 gpurun_out/full/pytest.log
