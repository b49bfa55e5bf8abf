export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
