"""namp_decoding_order against torch.argsort on random rows (all lengths 1..300, then up to 8192), with masks, ties and NaNs; and its time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import hip
L_ = hip.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
bad = 0
for L in list(range(1, 300)) + [511, 512, 513, 1000, 1024, 2047, 2400, 4096, 5000, 8192]:
    B = 3
    mask = (torch.rand(B, L, generator=g) > 0.1).float().to(dev)
    cm = (torch.rand(B, L, generator=g) > 0.3).float().to(dev)
    r = torch.randn(B, L, generator=g).to(dev)
    order = torch.empty(B, L, dtype=torch.int64, device=dev); rank = torch.empty(B, L, dtype=torch.int32, device=dev)
    hip.check(L_.namp_decoding_order(mask.data_ptr(), cm.data_ptr(), r.data_ptr(), order.data_ptr(), None, rank.data_ptr(), B, B, L, hip.current_stream()))
    key = (mask * cm + 0.0001) * r.abs()
    ref = torch.argsort(key, stable=True)
    if not torch.equal(order, ref): bad += 1; print("MISMATCH at L =", L)
    inv = torch.empty_like(ref); inv.scatter_(1, ref, torch.arange(L, device=dev).expand(B, L))
    assert torch.equal(rank.long(), inv), L
print("mismatches:", bad)
for (B, L) in ((1, 97), (1, 1000), (13, 2400), (30, 389)):
    mask = torch.ones(B, L, device=dev); r = torch.randn(B, L, device=dev)
    order = torch.empty(B, L, dtype=torch.int64, device=dev); rank = torch.empty(B, L, dtype=torch.int32, device=dev)
    f = lambda: hip.check(L_.namp_decoding_order(mask.data_ptr(), None, r.data_ptr(), order.data_ptr(), None, rank.data_ptr(), B, B, L, hip.current_stream()))
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} L={L}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch")
