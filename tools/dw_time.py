#!/usr/bin/env python
"""Time one message-stage backward launch at the cfg5 size (B=16, N=1500, K=48: 1.152 M edge rows) — the persistent launch that owns its
weight gradients (namp_train_edge_bwd_dw) and, for comparison, the round-3 form (namp_train_edge_bwd + the two row contractions).

    [NAMP_LIB_PATH=tools/_variants/x.so] python tools/dw_time.py [--prec 1|2] [--mode 0|1] [--reps 10] [--old]

Used with the DW_EXP_* ablation builds of csrc/namp_train_dw.h (tools/build_variants.sh: only namp_train.hip changes).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import hip, train          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", type=int, default=2)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--N", type=int, default=1500)
    ap.add_argument("--K", type=int, default=48)
    ap.add_argument("--old", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="DW_EXP_STAMPS build: print workgroup 0's phase timeline (s_memtime deltas)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = hip.lib()
    B, N, K, H = a.B, a.N, a.K, 128
    E = B * N * K
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev)
    h_E, Pa, Pj0, Pj1 = rn(B, N, K, H), rn(B, N, H), rn(B, N, H), rn(B, N, H)
    W1b, W2, b2 = rn(H, H, sc=0.08), rn(H, H, sc=0.1), rn(H, sc=0.1)
    # kNN-like neighbour lists: mostly sequence-local indices
    base = torch.arange(N)[None, :, None] + torch.randint(-40, 40, (B, N, K), generator=g)
    E_idx = base.clamp(0, N - 1).to(torch.int32).to(dev).contiguous()
    mask = torch.ones(B, N, dtype=torch.int32, device=dev)
    rank = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(torch.int32).to(dev).contiguous()
    gnode, gpass = rn(B * N, H), rn(E, H)
    img = lambda W, t=False: train._image(W, a.prec, transposed=t)
    i1, i2, i2t, i1t = img(W1b), img(W2), img(W2, True), img(W1b, True)
    rdt = torch.bfloat16 if a.prec == 2 else torch.float32
    Ep = L.namp_train_edge_bwd_dw_rows(B, N, K)
    G1 = torch.empty(Ep, H, device=dev, dtype=rdt)
    g_hE = torch.empty(Ep, H, device=dev)
    g_Pa = torch.empty(Ep // 16, H, device=dev)
    n = L.namp_train_edge_bwd_dw_groups(B, N, K)
    dWp, dbp = torch.empty(n, 2, H, H, device=dev), torch.empty(n, H, device=dev)
    flags = a.prec | 4 | 8
    s = hip.current_stream()
    m32, r32 = (mask.data_ptr() if a.mode == 0 else None), (rank.data_ptr() if a.mode == 1 else None)
    pj1 = Pj1.data_ptr() if a.mode == 1 else None
    if a.old:
        A1, G2 = (torch.empty(E, H, device=dev, dtype=rdt) for _ in range(2))

        def run():
            hip.check(L.namp_train_edge_bwd(a.mode, h_E.data_ptr(), E_idx.data_ptr(), m32, None, r32, Pa.data_ptr(), Pj0.data_ptr(), pj1,
                                            i1.data_ptr(), i2.data_ptr(), None, i2t.data_ptr(), i1t.data_ptr(), b2.data_ptr(), gnode.data_ptr(),
                                            A1.data_ptr(), None, G1.data_ptr(), G2.data_ptr(), None, g_hE.data_ptr(), gpass.data_ptr(),
                                            g_Pa.data_ptr(), None, None, None, None, flags, B, N, K, s), "bwd")
            train._wgrad_many([(G2, A1, True), (G1, h_E.view(E, H), False)], x3=a.prec)
    else:
        def run():
            hip.check(L.namp_train_edge_bwd_dw(a.mode, h_E.data_ptr(), E_idx.data_ptr(), m32, None, r32, Pa.data_ptr(), Pj0.data_ptr(), pj1,
                                               i1.data_ptr(), i2.data_ptr(), i2t.data_ptr(), i1t.data_ptr(), b2.data_ptr(), gnode.data_ptr(),
                                               G1.data_ptr(), g_hE.data_ptr(), gpass.data_ptr(), g_Pa.data_ptr(), dWp.data_ptr(), dbp.data_ptr(),
                                               flags, B, N, K, s), "bwd_dw")
    if a.stamps:
        st = torch.zeros(8 * 4 * 32, dtype=torch.int64, device=dev)
        os.environ["NAMP_DW_STAMPS"] = hex(st.data_ptr())
        run(); torch.cuda.synchronize()
        st.zero_()
        run(); torch.cuda.synchronize()
        t = st.view(8, 4, 32).cpu().numpy()
        names = ["top", "loads issued", "gemm1", "gelu1", "gemm2", "gelu2+g2", "barrier1", "stage1", "barrier2", "contract1", "gemm3", "g1+loads",
                 "barrier3", "stage2", "barrier4", "stores", "contract2", "gemm4", "end"]
        for w in range(4):
            print(f"wave {w}: deltas (s_memtime ticks = shader cycles) per phase, rounds 2..6")
            for r in range(2, 7):
                row = t[r, w, :19]
                print("   round", r, " ".join(f"{names[i + 1]}={int(row[i + 1] - row[i])}" for i in range(18)), "| total", int(row[18] - row[0]))
        return
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    rounds = (E + 63) // 64
    print(f"{os.path.basename(os.environ.get('NAMP_LIB_PATH', 'base'))} prec={a.prec} mode={a.mode} {'old' if a.old else 'dw'}: {ms:.4f} ms per launch"
          + ("" if a.old else f" = {ms * 1e3 / (rounds / n):.2f} us per 64-row round ({n} workgroups)"))


if __name__ == "__main__":
    main()
