"""Throughput of namp_encdec_fwd against batch size at N residues per complex (default precision), to place the
fused-tail / unfused and persistent-kernel thresholds.   python tools/batch_sweep.py [N] [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import hip, spec, synth
from na_mpnn_amd.pack import PackedWeights
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
L = hip.lib()
w = synth.make_weights(0)
P = PackedWeights({k: torch.from_numpy(v).to(dev) for k, v in w.items()}, 3, 3, spec.VOCAB, dev)
for B in (1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64):
    g = synth.make_graph(seed=5, batch=B, n=N, k=K)
    d = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    order = torch.argsort((d["mask"] * d["chain_mask"] + 0.0001) * torch.abs(d["randn"]))
    rank = torch.empty_like(order); rank.scatter_(1, order, torch.arange(N, device=dev).expand(B, -1)); rank = rank.to(torch.int32)
    hV = torch.empty(B, N, 128, device=dev); hE = torch.empty(B, N, K, 128, device=dev); logp = torch.empty(B, N, 33, device=dev)
    ws = torch.empty(2 * L.namp_workspace_bytes(B, B, N, K), dtype=torch.uint8, device=dev)
    def f():
        hip.check(L.namp_encdec_fwd(P.model(), d["V"].data_ptr(), d["E"].data_ptr(), d["E_idx"].data_ptr(), d["mask"].data_ptr(),
                                    d["S"].data_ptr(), rank.data_ptr(), hV.data_ptr(), hE.data_ptr(), logp.data_ptr(), None,
                                    ws.data_ptr(), ws.numel(), B, N, K, hip.current_stream()))
    for _ in range(3): f()
    reps = max(3, 200 // B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"B={B:3d} N={N} K={K}: {ms:8.3f} ms  {B * N / ms / 1e3:7.2f} M residues/s", flush=True)
