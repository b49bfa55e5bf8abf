# Round 4, session 10: rest of the GPU suite (from test_gpu_bench on) + clock poller check
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4_s10
python - <<'PY'
import torch, glob
try:
    print("torch.cuda.clock_rate:", torch.cuda.clock_rate(0))
except Exception as e:
    print("clock_rate failed:", type(e).__name__, str(e)[:100])
print(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")[:1]:
    print(open(f).read())
PY
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r4_s10/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_s10/pytest.log
tail -12 gpurun_out/r4_s10/pytest.log
