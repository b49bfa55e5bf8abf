# Round 3, session 3: GPU tests touched so far
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s3
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -x -q -k "ctor_variants or cfg3_sized or cli or include_pred or persistent" -s 2>&1 | tail -15
