"""Does data that still sits in the 256 MiB Infinity Cache make the row contractions (namp_train_wgrad_multi) or a plain copy faster?
    python tools/mall_probe.py
Times (a) torch's device-to-device copy and (b) the two-pair x3 row contraction at working sets from 32 MiB to 2.4 GiB, each repeated
back-to-back on the SAME buffers (so a working set below the cache size is re-read from it)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import train

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print("copy (read + write), bytes moved per second:")
for mb in (16, 32, 64, 96, 128, 256, 512, 2048):
    a = torch.empty(mb << 18, device=dev); b = torch.empty_like(a)          # mb MiB each
    t = timed(lambda: b.copy_(a), 50 if mb <= 256 else 10)
    print(f"  {mb:5d} MiB -> {mb:5d} MiB: {2 * mb / 1024 / t:8.1f} GiB/s  ({t * 1e6:.1f} us)")
    del a, b
print("row contraction, 2 pairs (G2^T A1, G1^T hE), split-bf16, fp32 rows:")
for rows in (36000, 72000, 144000, 288000, 576000, 1152000):
    t4 = [torch.randn(rows, 128, device=dev) for _ in range(4)]
    fn = lambda: train._wgrad_many([(t4[0], t4[1], True), (t4[2], t4[3], False)], x3=1)
    t = timed(fn, 20)
    mb = 4 * rows * 512 / 2**20
    print(f"  {rows:8d} rows ({mb:7.1f} MiB read): {t * 1e6:8.1f} us = {t / rows * 1e9:.3f} ns per row, {mb / 1024 / t:7.1f} GiB/s")
