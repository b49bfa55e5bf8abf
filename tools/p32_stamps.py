"""Per-wave time line of the re-sequenced bf16-storage edge launches (build with -DP32_STAMPS; NAMP_LIB_PATH selects it): cycles per tile between
the stamp points of namp_bf16p.h, averaged over the tiles of the eight waves of workgroup 0, for each launch kind of the cfg3 step."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, bench
from na_mpnn_amd import hip
L = hip.lib()
dbg = C.CDLL(os.environ["NAMP_LIB_PATH"]).namp_debug_p32_stamps
dev = torch.device("cuda:0")
args = argparse.Namespace(no_pmc=True, no_cpu_baseline=True, verbose=False, detail_out=None, min_seconds=1.0)
L.namp_set_bf16p(7)
o, r = bench.encdec_bench(args, dev, 0, 1, None, "cfg3", "bf16", 3, 2, profile_steps=3)
torch.cuda.synchronize()
buf = np.zeros((4, 8, 12), dtype=np.uint64)
rc = dbg(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
names = ["top..", "rows awaited", "requests+meta", "slots q1", "slots q2", "slots q3", "slots q4", "last epilogue", "LN / tail", "stores"]
for k, nm in ((0, "enc message"), (1, "dec message"), (2, "edge update"), (3, "enc message + embedding")):
    n = buf[k, :, 10].astype(np.float64)
    if n.sum() == 0: continue
    per = buf[k, :, :10].astype(np.float64) / n[:, None]
    print(f"== {nm}: {n.mean():.0f} tiles per wave; s_memtime ticks per tile (mean over 8 waves; 100 MHz ticks x ~20 = cycles): total {per.sum(1).mean():.1f}")
    for i, nn in enumerate(names):
        print(f"   {nn:16s} {per[:, i].mean():8.2f}  (min {per[:, i].min():.2f} max {per[:, i].max():.2f})")
