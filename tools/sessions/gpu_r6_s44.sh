R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
echo "== shared distances"; timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0'
echo "== HEAD"; NAMP_LIB_PATH=$R/tools/_variants/head.so timeout 600 python tools/feat_pipe_ab.py 2>&1 | grep -v amdgpu | grep x3 | awk 'NR%2==0'
done
timeout 600 python tools/feat_dump.py /tmp/a.npz 2>&1 | grep -v amdgpu
NAMP_LIB_PATH=$R/tools/_variants/head.so timeout 600 python tools/feat_dump.py /tmp/b.npz 2>&1 | grep -v amdgpu
python tools/feat_dump.py cmp /tmp/a.npz /tmp/b.npz | grep x3
python - <<'PY'
import numpy as np
for f in ("/tmp/a.npz", "/tmp/b.npz"):
    a = np.load(f)
    for mask in (11, 43):
        print(f, mask, "x3 vs fp32: max|dE| = %.3e  max|dh_E| = %.3e" % (np.abs(a[f"E_x3_{mask}"] - a[f"E_fp32_{mask}"]).max(), np.abs(a[f"hE_x3_{mask}"] - a[f"hE_fp32_{mask}"]).max()))
PY
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py tests/test_gpu_train.py -x -q 2>&1 | tail -3
