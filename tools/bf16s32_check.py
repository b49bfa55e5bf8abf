"""Parity + timing of the bf16-storage message launch (edge_mlp_bf16s32_kernel, namp_bf16s_message) against a torch restatement.
    python tools/bf16s32_check.py [B] [N] [K]
The 16x16x32 kernel this replaced ran the same inputs (rows in its own [s][g][j] order) in 538 / 546 us (encoder / decoder message,
B=64 N=1000 K=48, random neighbour indices) against 460 / 463 us here; inside the cfg3 step (real kNN indices) the difference is smaller:
profiles/r03e_bf16s32.md."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import hip


def frag_perm():
    """position -> channel of fragment order B (include/namp.h, namp_bf16s_message)"""
    perm = torch.empty(128, dtype=torch.long)
    for p in range(128):
        s, hk, j = p // 16, (p // 8) % 2, p % 8
        perm[p] = 16 * s + 8 * (j >> 2) + 4 * hk + (j & 3)
    assert sorted(perm.tolist()) == list(range(128))
    return perm


def case(B, N, K, dev, seed=1, time_it=False):
    """Runs both message modes; returns {mode: (max |dS| against torch on a residue subset, max |S|, us per launch or None)}."""
    L = hip.lib()
    G, E, tpn = B * N, B * N * K, (K + 15) // 16
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    perm = frag_perm().to(dev)
    hE = rn(E, 128).bfloat16(); Pa = rn(G, 128).bfloat16(); Pj0 = rn(G, 128).bfloat16(); Pj1 = rn(G, 128).bfloat16()
    W1 = (rn(128, 128) * 0.09).contiguous(); W2 = (rn(128, 128) * 0.09).contiguous(); b2 = (rn(128) * 0.1).contiguous()
    idx = torch.randint(0, N, (B, N, K), generator=g).to(dev).to(torch.int32)
    mask = (torch.rand(G, generator=g) > 0.05).to(dev).to(torch.int32)
    rank = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(dev).to(torch.int32).contiguous()
    i1 = torch.empty(128 * 128, dtype=torch.bfloat16, device=dev); i2 = torch.empty_like(i1)
    st = hip.current_stream()
    hip.check(L.namp_pack_image_bf16_32(W1.data_ptr(), 128, 0, i1.data_ptr(), st)); hip.check(L.namp_pack_image_bf16_32(W2.data_ptr(), 128, 0, i2.data_ptr(), st))
    t = [x[:, perm].contiguous() for x in (hE, Pa, Pj0, Pj1)]
    out = {}
    for mode in (0, 1):
        part = torch.full((G * tpn * 129 + 4,), float("nan"), device=dev)
        run = lambda: hip.check(L.namp_bf16s_message(mode, t[0].data_ptr(), idx.data_ptr(), mask.data_ptr(), rank.data_ptr(), t[1].data_ptr(),
                                                     t[2].data_ptr(), t[3].data_ptr(), i1.data_ptr(), i2.data_ptr(), b2.data_ptr(),
                                                     part.data_ptr(), B, B, N, K, st), "bf16s_message")
        run(); torch.cuda.synchronize()
        us = None
        if time_it:
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
        S = part[:G * tpn * 128].view(G, tpn, 128).sum(1)
        ws = part[G * tpn * 128:G * tpn * 129].view(G, tpn).sum(1)
        sub = torch.arange(0, G, max(1, G // 512), device=dev)
        sub = torch.unique(torch.cat([sub, torch.tensor([G - 1], device=dev)]))                 # the last residue: the odd trailing tile
        bq = sub // N
        jg = bq[:, None] * N + idx.view(G, K)[sub].long()
        x = hE.float().view(G, K, 128)[sub]
        if mode == 0:
            pj = Pj0.float()[jg]; w = (mask[sub][:, None] * mask[jg]).float() / 30.0
        else:
            bw = rank.view(-1)[jg] < rank.view(-1)[sub][:, None]
            pj = torch.where(bw[..., None], Pj0.float()[jg], Pj1.float()[jg]); w = torch.full(jg.shape, 1.0 / 30.0, device=dev)
        z1 = x @ W1.bfloat16().float().t() + Pa.float()[sub][:, None, :] + pj
        a1 = torch.nn.functional.gelu(z1).bfloat16().float()
        a2 = torch.nn.functional.gelu(a1 @ W2.bfloat16().float().t() + b2)
        ref = (w[..., None] * a2).sum(1)
        dw = float((ws[sub] - w.sum(1)).abs().max())
        out[mode] = (float((S[sub] - ref).abs().max()), float(ref.abs().max()), us, dw)
    return out


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    B, N, K = (int(x) for x in (sys.argv[1:4] + ["64", "1000", "48"][len(sys.argv) - 1:]))
    for mode, (err, smax, us, dw) in case(B, N, K, torch.device("cuda:0"), time_it=True).items():
        print(f"mode {mode}: {us:.1f} us per launch; vs torch (exact-erf GELU, fp32 accumulate): max |dS| = {err:.3e} (|S| max {smax:.2f}), weight sums off by {dw:.1e}")
