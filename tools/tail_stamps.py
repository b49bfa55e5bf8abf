"""Phase attribution of the residue tail inside the fused cfg2 launches (NAMP_ABL_STAMPS builds; workgroup 0, 10 ns ticks):
    tools/build_variants.sh stamps:-DNAMP_ABL_STAMPS stamps_valu:-DNAMP_ABL_STAMPS,-DNAMP_TAIL_VALU
    NAMP_LIB_PATH=tools/_variants/stamps.so python tools/tail_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
L = C.CDLL(os.environ["NAMP_LIB_PATH"])
names = {7: "hoisted layer 3 + x = h_V + m", 8: "LayerNorm 1 + x to LDS", 9: "W_in + GELU", 10: "W_out partials", 11: "LayerNorm 2 + h_V' out",
         12: "output head + projections"}
for prec in ("fp32", "x3"):
    step = bench.Runner(dev, 1, 1000, 48, 0, precision=prec).step
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    L.namp_debug_stamps(buf, 1)
    reps = 20
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    L.namp_debug_stamps(buf, 1)
    print(f"cfg2 {prec}: us per forward (6 tails), workgroup 0")
    tot = 0.0
    for i, nm in names.items():
        v = buf[i] * 0.01 / reps
        tot += v
        print(f"   {nm:34s} {v:8.2f}   ({v / 6:5.2f} per launch)")
    print(f"   {'total':34s} {tot:8.2f}   ({tot / 6:5.2f} per launch)")
