"""Per-wave time line of the sampler step from the NAMP_ABL_WSTAMPS build (tools/build_variants.sh wstamps:-DNAMP_ABL_WSTAMPS):
    NAMP_LIB_PATH=tools/_variants/wstamps.so python tools/sample_wstamps.py
Every wave of workgroup 0 logs (slot, s_memtime) without synchronising; this prints, per phase (the interval ENDING at a slot), the mean over
levels and layers of the per-wave durations (mean over waves / slowest wave), in us at the polled clock."""
import ctypes as C, os, sys
from collections import defaultdict
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
torch.manual_seed(0)
w = synth.make_weights(0)
L = C.CDLL(os.environ["NAMP_LIB_PATH"])
NAMES = {0: "step start (after the previous step's last barrier)", 2: "layer argument block read", 1: "index chain + row gather (loads landed, z1 formed)",
         3: "product 2 (gelu, W2)", 20: "product 3 (gelu, W3) + K-sum", 4: "  barrier wait", 21: "next images copied", 7: "K-sums + h_V read (tail entry)",
         8: "LayerNorm 1", 9: "W_in + GELU", 22: "W_out partials", 10: "  barrier wait", 13: "partial sums read", 14: "LayerNorm 2", 11: "mask, h_V' out",
         23: "projections", 5: "  barrier wait", 24: "head + draw", 6: "  barrier wait"}
n, k, bs = 97, 32, 1
m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                polytype_to_int=spec.polytype_to_int())
m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()}); m = m.to(dev).eval()
cx = synth.make_complex(seed=3, n=n)
fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k_, v in cx.items()}
fd.update({"batch_size": bs, "temperature": 0.1, "bias": torch.zeros(1, n, 33, device=dev), "symmetry_residues": [[]],
           "symmetry_weights": [[]], "randn": torch.randn(bs, n, device=dev)})
out = m.sample(fd); torch.cuda.synchronize()
cnt = (C.c_int * 8)()
NEV = 8192
log = (C.c_longlong * (8 * NEV * 2))()
assert L.namp_debug_wstamps(cnt, log, 1) == NEV
out = m.sample(fd); torch.cuda.synchronize()
L.namp_debug_wstamps(cnt, log, 1)
lv = int(out["levels"])
arr = np.frombuffer(log, dtype=np.int64).reshape(8, NEV, 2)
dur = defaultdict(lambda: [[] for _ in range(8)])
span = []
for wv in range(8):
    ev = arr[wv, :cnt[wv]]
    for i in range(1, len(ev)):
        dur[int(ev[i, 0])][wv].append(int(ev[i, 1] - ev[i - 1, 1]))
    span.append(int(ev[-1, 1] - ev[0, 1]))
import time
t0 = time.perf_counter(); out = m.sample(fd); torch.cuda.synchronize(); call_ms = (time.perf_counter() - t0) * 1e3
clk = float(os.environ.get("NAMP_CLK_MHZ", "0")) or span[0] / max((call_ms - 0.7) * 1e3, 1.0)   # ticks per us: calibrated on the call itself (~0.7 ms outside the walk)
print(f"call {call_ms:.2f} ms -> {clk:.0f} s_memtime ticks per us")
print(f"N={n} K={k}: {lv} levels, {cnt[0]} events per wave, wave-0 span {span[0] / clk:.1f} us = {span[0] / clk / lv:.2f} us per level")
print(f"{'phase (interval ending at the stamp)':58s} {'count/level':>11s} {'mean us':>8s} {'slowest wave':>12s} {'us/level':>9s}")
tot = 0.0
for slot, name in NAMES.items():
    per_wave = [np.mean(d) / clk if d else 0.0 for d in dur[slot]]
    c = len(dur[slot][0]) / lv
    mean_ = float(np.mean(per_wave)); mx = float(np.max(per_wave))
    tot += mean_ * c
    print(f"{name:58s} {c:11.2f} {mean_:8.2f} {mx:12.2f} {mean_ * c:9.2f}")
print(f"{'sum':58s} {'':11s} {'':8s} {'':12s} {tot:9.2f}")
