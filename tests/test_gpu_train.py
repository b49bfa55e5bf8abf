"""GPU tests of the training path: the HIP backward kernels against fp64 torch autograd of the
dense formulas, and one full training step against the golden the real reference produced (G7)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from na_mpnn_amd import spec, synth, train
from na_mpnn_amd.model import ProteinMPNN
from oracle import cpu_ref
from test_oracle_golden import g7_inputs, g7b_ppm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_model(weights_np, k, dropout=0.0):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, dropout=dropout, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in weights_np.items()})
    return m.to(DEV)


@pytest.mark.parametrize("mode,B,N,K", [(0, 2, 37, 20), (1, 2, 37, 20), (2, 1, 50, 48), (0, 1, 64, 48), (1, 3, 21, 16)])
def test_edge_mlp_backward_matches_autograd(mode, B, N, K):
    """namp_train_edge_fwd / _bwd / _wgrad vs fp64 autograd of z3 = W3 gelu(W2 gelu(W1b h_E + Pa_i + Pj_j) + b2) + b3."""
    g = torch.Generator(device="cpu").manual_seed(100 * mode + K)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    h_E, Pa, Pj0, Pj1 = rn(B, N, K, 128), rn(B, N, 128), rn(B, N, 128), rn(B, N, 128)
    W1 = rn(128, 384, sc=0.08)
    W2, W3, b2, b3 = rn(128, 128, sc=0.1), rn(128, 128, sc=0.1), rn(128, sc=0.1), rn(128, sc=0.1)
    E_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(DEV)
    mask = (torch.rand(B, N, generator=g) > 0.15).to(DEV)
    rank = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(DEV)
    R = rn(B, N, K, 128) if mode == 2 else rn(B, N, 128)
    leaves = [h_E, Pa, Pj0, Pj1, W1, W2, b2, W3, b3]

    def dense(hE, pa, pj0, pj1, w1, w2, bb2, w3, bb3):
        bidx = torch.arange(B, device=DEV)[:, None, None]
        if mode == 1:
            bw = (rank[bidx, E_idx] < rank[:, :, None]).unsqueeze(-1)
            pj = torch.where(bw, pj0[bidx, E_idx], pj1[bidx, E_idx])
        else:
            pj = pj0[bidx, E_idx]
        z1 = hE @ w1[:, 128:256].t() + pa[:, :, None] + pj
        z3 = F.gelu(F.gelu(z1) @ w2.t() + bb2) @ w3.t() + bb3
        if mode == 2:
            return z3
        wgt = (mask[:, :, None] & mask[bidx, E_idx]).to(z3.dtype) if mode == 0 else torch.ones_like(z3[..., 0])
        return (wgt.unsqueeze(-1) * z3).sum(2) / 30.0

    with torch.enable_grad():
        ref_in = [t.double().requires_grad_(True) for t in leaves]
        ref_out = dense(*ref_in)
        # message modes hand h_E through to its next consumer: a second gradient arrives on it and must be ADDED in place
        R2 = rn(B, N, K, 128) if mode != 2 else None
        ((ref_out * R.double()).sum() + ((ref_in[0] * R2.double()).sum() if mode != 2 else 0.0)).backward()
        ours_in = [t.clone().requires_grad_(True) for t in leaves]
        hE, pa, pj0, pj1, w1, w2, bb2, w3, bb3 = ours_in
        out = train._EdgeMLP.apply(mode, hE, pa, pj0, pj1 if mode == 1 else None, w1[:, 128:256], w2, bb2, w3, bb3,
                                   E_idx.to(torch.int32).contiguous(), mask.to(torch.int32).contiguous() if mode == 0 else None,
                                   None, rank.to(torch.int32).contiguous() if mode == 1 else None)
        if mode != 2:
            out, h_pass = out
            assert h_pass.data_ptr() == hE.data_ptr()
            ((out * R).sum() + (h_pass * R2).sum()).backward()
        else:
            (out * R).sum().backward()
    assert rel(out, ref_out) < 2e-5
    names = ["h_E", "Pa", "Pj0", "Pj1", "W1", "W2", "b2", "W3", "b3"]
    for name, a, b in zip(names, ours_in, ref_in):
        if name == "Pj1" and mode != 1:
            continue
        assert a.grad is not None, name
        assert rel(a.grad, b.grad) < 5e-5, (name, rel(a.grad, b.grad))


# (one case per precision: these compare two HIP paths; the fp64 comparison of test_on_chip_backward_matches_fp64_autograd supersedes them)
@pytest.mark.parametrize("prec,mode,B,N,K", [(1, 0, 1, 333, 30), (2, 1, 1, 40, 16)])
def test_edge_mlp_backward_on_chip_weight_gradients(mode, B, N, K, prec, monkeypatch):
    """The persistent message backward that contracts (G2, A1) and (G1, h_E) on chip (csrc/namp_train_dw.h) against the round-3 form
    (row tensors to HBM + row-contraction launches) on the same inputs, at sizes where a workgroup walks several rounds (67,200 rows =
    1,050 rounds of 64 over <= 256 workgroups), with a ragged last round, K % 16 != 0 (atomic dL/dPa path) and a pass-through
    gradient.  Split-bf16 products: every gradient within 2e-5 of the other form (relative to its largest entry); bf16 products:
    within 1.5 % (both forms round the same operands to bf16; the summation orders differ)."""
    g = torch.Generator(device="cpu").manual_seed(7 + 10 * mode + K)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    h_E, Pa, Pj0, Pj1 = rn(B, N, K, 128), rn(B, N, 128), rn(B, N, 128), rn(B, N, 128)
    W1b, W2, W3 = rn(128, 128, sc=0.08), rn(128, 128, sc=0.1), rn(128, 128, sc=0.1)
    b2, b3 = rn(128, sc=0.1), rn(128, sc=0.1)
    E_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(DEV).to(torch.int32).contiguous()
    mask = (torch.rand(B, N, generator=g) > 0.15).to(DEV).to(torch.int32).contiguous()
    rank = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(DEV).to(torch.int32).contiguous()
    R, R2 = rn(B, N, 128), rn(B, N, K, 128)
    leaves = [h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, b3]
    monkeypatch.setattr(train, "X3", prec)

    def run(on_chip):
        monkeypatch.setattr(train, "DW_ONCHIP", on_chip)
        with torch.enable_grad():
            ins = [t.clone().requires_grad_(True) for t in leaves]
            hE, pa, pj0, pj1, w1b, w2, bb2, w3, bb3 = ins
            out, h_pass = train._EdgeMLP.apply(mode, hE, pa, pj0, pj1 if mode == 1 else None, w1b, w2, bb2, w3, bb3, E_idx,
                                               mask if mode == 0 else None, None, rank if mode == 1 else None)
            ((out * R).sum() + (h_pass * R2).sum()).backward()
        torch.cuda.synchronize()
        return out.detach(), [t.grad for t in ins]

    out_a, ga = run(False)
    out_b, gb = run(True)
    assert torch.equal(out_a, out_b)
    bar = 2e-5 if prec == 1 else 1.5e-2
    worst = {}
    for name, a, b in zip(["h_E", "Pa", "Pj0", "Pj1", "W1b", "W2", "b2", "W3", "b3"], ga, gb):
        if name == "Pj1" and mode != 1:
            continue
        assert a is not None and b is not None and torch.isfinite(b).all(), name
        worst[name] = rel(b, a)
        assert worst[name] < bar, (name, worst[name])
    print(f"on-chip dW vs row tensors (mode {mode}, prec {prec}, {B}x{N}x{K}):", {k_: f"{v:.1e}" for k_, v in worst.items()})


@pytest.mark.parametrize("B,N,K,p", [(1, 333, 30, 0.1)])
def test_edge_update_backward_on_chip_weight_gradients(B, N, K, p, monkeypatch):
    """Mixed precision: the edge update's two-launch backward that owns all three weight gradients (csrc/namp_train_eu.h) against the
    round-3 form on the same inputs — several rounds per workgroup, a ragged last round, K % 16 != 0 (atomic dL/dPa path), dropout on and
    off.  Both forms round the same operands to bf16; summation orders (and the GELU polynomial) differ: every gradient within 1.5 %
    of the other form, relative to its largest entry.  (Against fp64 autograd directly: test_on_chip_backward_matches_fp64_autograd.)"""
    g = torch.Generator(device="cpu").manual_seed(23 + K)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    h_E, Pa, Pc = rn(B, N, K, 128), rn(B, N, 128), rn(B, N, 128)
    W1b, W2, W3 = rn(128, 128, sc=0.08), rn(128, 128, sc=0.1), rn(128, 128, sc=0.1)
    b2, b3, lw, lb = rn(128, sc=0.1), rn(128, sc=0.1), 1 + rn(128, sc=0.1), rn(128, sc=0.1)
    E32 = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(DEV).to(torch.int32).contiguous()
    R = rn(B, N, K, 128)
    leaves = [h_E, Pa, Pc, W1b, W2, b2, W3, b3, lw, lb]
    names = ["h_E", "Pa", "Pc", "W1b", "W2", "b2", "W3", "b3", "ln_w", "ln_b"]
    monkeypatch.setattr(train, "X3", 2)

    def run(on_chip):
        monkeypatch.setattr(train, "DW_ONCHIP_EDGE", on_chip)
        with torch.enable_grad():
            ins = [t.clone().requires_grad_(True) for t in leaves]
            out = train._EdgeUpdate.apply(*ins, E32, p, 4321)
            (out * R).sum().backward()
        torch.cuda.synchronize()
        return out.detach(), [t.grad for t in ins]

    out_a, ga = run(False)
    out_b, gb = run(True)
    assert torch.equal(out_a, out_b)
    worst = {}
    for name, a, b in zip(names, ga, gb):
        assert a is not None and b is not None and torch.isfinite(b).all(), name
        worst[name] = rel(b, a)
        assert worst[name] < 1.5e-2, (name, worst[name])
    print(f"edge update, on-chip dW vs row tensors ({B}x{N}x{K}, p={p}):", {k_: f"{v:.1e}" for k_, v in worst.items()})


def _mix32(x):
    """mix32 of csrc/namp_device.h on int64 tensors holding uint32 values."""
    M = 0xFFFFFFFF
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & M
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & M
    return x ^ (x >> 16)


def hash_dropout_factor(seed, E, p, dev):
    """The training path's counter-based dropout mask (drop_row_key / drop_factor, csrc/namp_device.h) restated in torch: [E,128] of
    0 or 1 / (1 - p).  Edge rows < 2^32, so the key's high-word term is mix32(0x9E3779B9)."""
    M = 0xFFFFFFFF
    e = torch.arange(E, device=dev, dtype=torch.int64)
    hi = _mix32(torch.full((1,), 0x9E3779B9, device=dev, dtype=torch.int64))
    key = _mix32((int(seed) & M) ^ e ^ hi)
    ch = torch.arange(128, device=dev, dtype=torch.int64)
    h = _mix32((key[:, None] + ch[None, :] * 0x9E3779B9) & M)
    thresh = int(float(p) * 4294967296.0)
    return (h >= thresh).double() / (1.0 - p)


@pytest.mark.parametrize("kind,prec", [("enc_msg", 1), ("enc_msg", 2), ("dec_msg", 1), ("dec_msg", 2), ("edge", 2)])
def test_on_chip_backward_matches_fp64_autograd(kind, prec, monkeypatch):
    """The weight-gradient-owning backward launches (message stages: csrc/namp_train_dw.h; edge update: csrc/namp_train_eu.h) against fp64
    torch autograd of the dense formulas DIRECTLY, at a size where every persistent workgroup walks several rounds and the cross-round
    accumulators matter (2 x 700 x 48 = 67,200 edge rows = 1,050 rounds of 64 over <= 256 workgroups; VERDICT r4 parity hole 1).  The edge
    update runs with dropout 0.1, its hash mask restated in torch.  Split-bf16 products: 5e-5 of the largest entry; bf16 products (operands
    rounded to bf16, degree-4 GELU): 3 %."""
    B, N, K, p, seed = 2, 700, 48, 0.1, 4321
    g = torch.Generator(device="cpu").manual_seed(99 + prec)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    h_E, Pa, Pj0, Pj1 = rn(B, N, K, 128), rn(B, N, 128), rn(B, N, 128), rn(B, N, 128)
    W1b, W2, W3 = rn(128, 128, sc=0.08), rn(128, 128, sc=0.1), rn(128, 128, sc=0.1)
    b2, b3, lw, lb = rn(128, sc=0.1), rn(128, sc=0.1), 1 + rn(128, sc=0.1), rn(128, sc=0.1)
    E_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(DEV)
    E32 = E_idx.to(torch.int32).contiguous()
    mask = (torch.rand(B, N, generator=g) > 0.15).to(DEV)
    rank = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(DEV)
    bidx = torch.arange(B, device=DEV)[:, None, None]
    monkeypatch.setattr(train, "X3", prec)
    monkeypatch.setattr(train, "DW_ONCHIP", True)
    monkeypatch.setattr(train, "DW_ONCHIP_EDGE", True)
    if kind == "edge":
        R = rn(B, N, K, 128)
        leaves = [h_E, Pa, Pj0, W1b, W2, b2, W3, b3, lw, lb]
        names = ["h_E", "Pa", "Pc", "W1b", "W2", "b2", "W3", "b3", "ln_w", "ln_b"]
        with torch.enable_grad():
            ins = [t.clone().requires_grad_(True) for t in leaves]
            out = train._EdgeUpdate.apply(*ins, E32, p, seed)
            (out * R).sum().backward()
            ref_in = [t.double().requires_grad_(True) for t in leaves]
            hE, pa, pc, w1, w2, bb2, w3, bb3, w_, b_ = ref_in
            z1 = hE @ w1.t() + pa[:, :, None] + pc[bidx, E_idx]
            z3 = F.gelu(F.gelu(z1) @ w2.t() + bb2) @ w3.t() + bb3
            drop = hash_dropout_factor(seed, B * N * K, p, DEV).view(B, N, K, 128)
            ref = F.layer_norm(hE + z3 * drop, (128,), w_, b_, 1e-5)
            (ref * R.double()).sum().backward()
    else:
        mode = 0 if kind == "enc_msg" else 1
        R, R2 = rn(B, N, 128), rn(B, N, K, 128)
        leaves = [h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, b3]
        names = ["h_E", "Pa", "Pj0", "Pj1", "W1b", "W2", "b2", "W3", "b3"]
        with torch.enable_grad():
            ins = [t.clone().requires_grad_(True) for t in leaves]
            hE, pa, pj0, pj1, w1, w2, bb2, w3, bb3 = ins
            out, h_pass = train._EdgeMLP.apply(mode, hE, pa, pj0, pj1 if mode == 1 else None, w1, w2, bb2, w3, bb3, E32,
                                               mask.to(torch.int32).contiguous() if mode == 0 else None, None,
                                               rank.to(torch.int32).contiguous() if mode == 1 else None)
            ((out * R).sum() + (h_pass * R2).sum()).backward()
            ref_in = [t.double().requires_grad_(True) for t in leaves]
            hE, pa, pj0, pj1, w1, w2, bb2, w3, bb3 = ref_in
            if mode == 1:
                bw = (rank[bidx, E_idx] < rank[:, :, None]).unsqueeze(-1)
                pj = torch.where(bw, pj0[bidx, E_idx], pj1[bidx, E_idx])
                wgt = torch.ones(B, N, K, device=DEV, dtype=torch.float64)
            else:
                pj = pj0[bidx, E_idx]
                wgt = (mask[:, :, None] & mask[bidx, E_idx]).double()
            z3 = F.gelu(F.gelu(hE @ w1.t() + pa[:, :, None] + pj) @ w2.t() + bb2) @ w3.t() + bb3
            ref = (wgt.unsqueeze(-1) * z3).sum(2) / 30.0
            ((ref * R.double()).sum() + (hE * R2.double()).sum()).backward()
    torch.cuda.synchronize()
    bar = 5e-5 if prec == 1 else 3e-2
    assert rel(out, ref) < (2e-5 if prec == 1 else 2e-2)
    worst = {}
    for name, a, b in zip(names, ins, ref_in):
        if name == "Pj1" and kind != "dec_msg":
            continue
        assert a.grad is not None and torch.isfinite(a.grad).all(), name
        worst[name] = rel(a.grad, b.grad)
    print(f"on-chip backward vs fp64 autograd ({kind}, prec {prec}):", {k_: f"{v:.1e}" for k_, v in worst.items()})
    for name, v in worst.items():
        assert v < bar, (name, v, worst)


@pytest.mark.parametrize("B,N,K", [(1, 50, 48), (3, 257, 30), (2, 1500, 48), (1, 6000, 16)])
def test_reverse_adjacency_equals_the_stable_sort(B, N, K):
    """namp_train_reverse_adjacency (counting sort + per-row rank sort on the device) against torch's stable argsort of the edges by target row;
    hub rows (every residue lists residue 0: a row with N incoming edges, > 64) included — up to the maximum complex size (N = 6000: the rank
    sort of such a row is one wave's N^2 / 64 compare steps; the bound asserted here is what that costs)."""
    import time
    g = torch.Generator().manual_seed(B * 1000 + N)
    E_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)])
    E_idx[:, :, 0] = 0                                           # a hub
    E32 = E_idx.to(torch.int32).to(DEV).contiguous()
    rev = train.ReverseAdjacency(E32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rev = train.ReverseAdjacency(E32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"reverse adjacency, B={B} N={N} K={K} with a hub of in-degree {N}: {dt * 1e3:.2f} ms")
    assert dt < 0.05
    jflat = (E_idx + (torch.arange(B) * N)[:, None, None]).view(-1)
    order = torch.argsort(jflat, stable=True).to(torch.int32)
    counts = torch.bincount(jflat, minlength=B * N)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32)
    assert torch.equal(rev.offsets.cpu(), offsets)
    assert torch.equal(rev.edges.cpu(), order)
    rank = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(torch.int32).to(DEV)
    sel = rev.decoder_sel(rank).cpu()
    r = rank.cpu().view(-1).long()
    ref = (r[jflat] < r.repeat_interleave(K)).to(torch.uint8)
    assert torch.equal(sel, ref)


def test_reduce_sum_segments_match_torch():
    """namp_reduce_sum: several segments of different shapes in one launch — [n][M] partials, per-tile rows [G][T][128] -> [G][128], a scalar
    (Mb = 1) segment and an odd-sized one on the scalar path — against fp64 sums; deterministic (two launches bit-identical)."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(37, 128, 128, generator=g).to(DEV)                 # [n][M]
    b = torch.randn(301, 3, 128, generator=g).to(DEV)                  # [G][T][128] over T
    c = torch.randn(1000, 3, generator=g).to(DEV)                      # [G][T] over T, Mb = 1
    d = torch.randn(5, 7, 33, generator=g).to(DEV)                     # [A][n][33] over n (33 % 4 != 0)
    segs = [train._seg0(a), (b, 301, 128, 3 * 128, 128, 3), (c, 1000, 1, 3, 1, 3), train._seg1(d)]
    o1 = train._reduce(*segs)
    o2 = train._reduce(*segs)
    refs = [a.double().sum(0).view(1, -1), b.double().sum(1), c.double().sum(1, keepdim=True), d.double().sum(1)]
    for x, y, r in zip(o1, o2, refs):
        assert torch.equal(x, y)
        assert rel(x, r) < 1e-6
    many = [train._seg0(torch.randn(4 + i, 64, generator=g).to(DEV)) for i in range(19)]      # > 16 segments: two launches
    for x, (t, *_r) in zip(train._reduce(*many), many):
        assert rel(x, t.double().sum(0).view(1, -1)) < 1e-6


def test_pack_images_equals_the_single_image_launches():
    """namp_pack_images (one launch over a descriptor table, transposition inside) against namp_pack_image_x3 / _bf16 / _x3_general on the same blocks,
    byte for byte — including column-sliced views and transposed blocks."""
    from na_mpnn_amd import hip
    g = torch.Generator().manual_seed(9)
    W1 = torch.randn(128, 384, generator=g).to(DEV)
    Win = torch.randn(512, 128, generator=g).to(DEV)
    plan = train._PackPlan()
    blocks = [(W1[:, 128:256], 1, False), (W1[:, 128:256], 1, True), (W1[:, :128], 2, False), (W1[:, 256:], 2, True),
              (Win, 3, False), (Win, 3, True)]
    for blk, kind, tr in blocks:
        plan.register(blk, kind, tr)
    tok = object()
    plan.begin_step(tok, torch.device(DEV))
    L = hip.lib()
    for blk, kind, tr in blocks:
        got = plan.lookup(blk, kind, tr, tok)
        assert got is not None
        src = blk.t().contiguous() if tr else blk
        ref = torch.empty(src.numel(), device=DEV)
        if kind == 1:
            hip.check(L.namp_pack_image_x3(src.data_ptr(), src.stride(0), 0, ref.data_ptr(), hip.current_stream()), "x3")
        elif kind == 2:
            hip.check(L.namp_pack_image_bf16(src.data_ptr(), src.stride(0), 0, ref.data_ptr(), hip.current_stream()), "bf16")
        else:
            hip.check(L.namp_pack_image_x3_general(src.data_ptr(), src.shape[1], 0, src.shape[0], src.shape[1], ref.data_ptr(), hip.current_stream()), "x3g")
        nbytes = got.numel() * 4
        assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8)[:nbytes]), (kind, tr)
    plan.begin_step(object(), torch.device(DEV))                        # entries nobody asked for during a step are forgotten
    plan.begin_step(object(), torch.device(DEV))
    assert not plan.entries


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("B,N,K", [(1, 40, 16), (2, 700, 48), (1, 333, 30)])
def test_edge_embed_tail_matches_fp64_autograd(B, N, K, prec, monkeypatch):
    """_EdgeEmbedTail: h_E = W_e LayerNorm(y) + b_e in one forward launch on the pre-LayerNorm rows, backward without the normalised rows in
    memory (W_e^T product + LayerNorm backward in one pass; dW_e from y and the rows' statistics) — against fp64 autograd of the dense formula
    at sizes where every persistent workgroup walks several tiles, and against the unfused pair of launches."""
    g = torch.Generator(device="cpu").manual_seed(17)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    leaves = [rn(B, N, K, 128, sc=2.0) + 0.5, 1 + rn(128, sc=0.2), rn(128, sc=0.2), rn(128, 128, sc=0.1), rn(128, sc=0.1)]
    R = rn(B, N, K, 128)
    monkeypatch.setattr(train, "X3", prec)
    with torch.enable_grad():
        ins = [t.clone().requires_grad_(True) for t in leaves]
        out = train._EdgeEmbedTail.apply(*ins)
        (out * R).sum().backward()
        ref_in = [t.double().requires_grad_(True) for t in leaves]
        ref = F.linear(F.layer_norm(ref_in[0], (128,), ref_in[1], ref_in[2], 1e-5), ref_in[3], ref_in[4])
        (ref * R.double()).sum().backward()
        old_in = [t.clone().requires_grad_(True) for t in leaves]
        old = train._EdgeLinear.apply(train._RowLayerNorm.apply(old_in[0], old_in[1], old_in[2]), old_in[3], old_in[4])
        (old * R).sum().backward()
    tol = 3e-5 if prec == 1 else 2e-2
    assert rel(out, ref) < tol, rel(out, ref)
    for name, a, b, c in zip(("y", "ln_w", "ln_b", "W_e", "b_e"), ins, ref_in, old_in):
        assert rel(a.grad, b.grad) < (1e-4 if prec == 1 else 3e-2), (name, rel(a.grad, b.grad))
        assert rel(a.grad, c.grad) < (2e-5 if prec == 1 else 2e-2), (name, rel(a.grad, c.grad))


def test_class_sums_and_weighted_column_sums_match_torch():
    """namp_train_class_sums (gradient of a few-row embedding lookup: per-class sums of [rows][128] by an index, out-of-range rows skipped) and
    namp_train_wcolsum (sum_rows g[row] * w[row]) against fp64; both deterministic."""
    from na_mpnn_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(9)
    for rows, ncls in ((1, 6), (1000, 6), (24001, 33), (70003, 64)):
        x = torch.randn(rows, 128, generator=g).to(DEV)
        idx = torch.randint(-1, ncls + 1, (rows,), generator=g).to(torch.int32).to(DEV)        # -1 and ncls: skipped
        w = torch.randn(rows, generator=g).to(DEV)
        n = L.namp_train_rows_groups(rows)
        outs = []
        for _ in range(2):
            part = torch.empty(n, ncls, 128, device=DEV)
            hip.check(L.namp_train_class_sums(x.data_ptr(), idx.data_ptr(), ncls, rows, part.data_ptr(), hip.current_stream()), "class_sums")
            pw = torch.empty(n, 128, device=DEV)
            hip.check(L.namp_train_wcolsum(x.data_ptr(), w.data_ptr(), rows, pw.data_ptr(), hip.current_stream()), "wcolsum")
            outs.append((train._reduce(train._seg0(part))[0].view(ncls, 128), train._reduce(train._seg0(pw))[0].view(128)))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        ok = (idx >= 0) & (idx < ncls)
        ref = torch.zeros(ncls, 128, dtype=torch.float64, device=DEV).index_add_(0, idx[ok].long(), x[ok].double())
        assert rel(outs[0][0], ref) < 1e-6
        assert rel(outs[0][1], (x.double() * w.double()[:, None]).sum(0)) < 1e-6


@pytest.mark.parametrize("prec", [1, 2])
def test_row_contraction_in_slices_adds_up(prec):
    """namp_train_wgrad_multi with bit 6 (add to the partials of an earlier launch) and an explicit chunk count: three launches over three row
    slices = one launch over all rows; against fp64 within the product precision."""
    g = torch.Generator().manual_seed(21)
    rows = 50000
    G1, A1, G2, A2 = (torch.randn(rows, 128, generator=g).to(DEV) for _ in range(4))
    cuts = [0, 20000, 37000, rows]
    sl = [[(G1[a:b], A1[a:b], True), (G2[a:b], A2[a:b], False)] for a, b in zip(cuts, cuts[1:])]
    whole = train._wgrad_many([(G1, A1, True), (G2, A2, False)], x3=prec)
    parts = train._wgrad_many(sl[0], x3=prec, more=iter(sl[1:]))
    tol = 3e-5 if prec == 1 else 2e-2             # split-bf16 products drop the mid x mid term (2^-16 relative per product); plain bf16: 2^-8
    for (dw, db), (dw2, db2), (G, A) in zip(whole, parts, ((G1, A1), (G2, A2))):
        assert rel(dw, dw2) < 1e-5, rel(dw, dw2)                           # same products, another order of the fp32 additions
        assert rel(dw, G.double().t() @ A.double()) < tol, rel(dw, G.double().t() @ A.double())
        if db is not None:
            assert rel(db, db2) < 1e-5 and rel(db, G.double().sum(0)) < 1e-5


def test_positional_features_and_gradients_match_torch():
    """namp_train_pos_features / namp_train_pos_grad against the torch expression of PositionalEncodings (na_model_utils.py:537-541) and fp64 autograd
    of E_pos . (g Wedge[:, :16])."""
    from na_mpnn_amd import hip
    B, N, K = 2, 211, 30
    g = torch.Generator().manual_seed(17)
    R = torch.cumsum(torch.randint(1, 4, (B, N), generator=g), 1).to(torch.int32)
    ch = (torch.arange(N)[None, :] // 60 + torch.arange(B)[:, None]).to(torch.int32)
    E_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)])
    W = torch.randn(16, 66, generator=g)
    b = torch.randn(16, generator=g)
    Wedge = torch.randn(128, 5200, generator=g) * 0.05
    gy = torch.randn(B, N, K, 128, generator=g)
    bidx = torch.arange(B)[:, None, None]
    off = R.long()[:, :, None] - R.long()[bidx, E_idx]
    same = (ch[:, :, None] == ch[bidx, E_idx]).long()
    d_ref = torch.clip(off + 32, 0, 64) * same + (1 - same) * 65
    Wd, bd = W.double().requires_grad_(), b.double().requires_grad_()
    with torch.enable_grad():
        E_pos_ref = Wd.t()[d_ref] + bd
        (E_pos_ref * (gy.double().view(-1, 128) @ Wedge.double()[:, :16]).view(B, N, K, 16)).sum().backward()
    L = hip.lib()
    dv = lambda t: t.to(DEV).contiguous()
    d32 = torch.empty(B, N, K, dtype=torch.int32, device=DEV)
    E_pos = torch.empty(B, N, K, 16, device=DEV)
    Rd, cd, Ed, Wv, bv, Wev, gv = dv(R), dv(ch), dv(E_idx.to(torch.int32)), dv(W), dv(b), dv(Wedge), dv(gy)
    hip.check(L.namp_train_pos_features(Rd.data_ptr(), cd.data_ptr(), Ed.data_ptr(), Wv.data_ptr(), bv.data_ptr(), d32.data_ptr(), E_pos.data_ptr(),
                                        B, N, K, hip.current_stream()), "pos_features")
    assert torch.equal(d32.cpu().long(), d_ref)
    assert rel(E_pos, E_pos_ref) < 1e-6
    E = B * N * K
    n = L.namp_train_pos_grad_groups(E)
    part = torch.empty(n, 67, 16, device=DEV)
    hip.check(L.namp_train_pos_grad(gv.data_ptr(), Wev.data_ptr(), Wev.stride(0), d32.data_ptr(), part.data_ptr(), E, hip.current_stream()), "pos_grad")
    tab = part.double().sum(0)
    assert rel(tab[:66].t(), Wd.grad) < 2e-5
    assert rel(tab[66], bd.grad) < 2e-5


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_edge_update_backward_in_slices_equals_one_launch(p, monkeypatch):
    """Split-bf16 mode walks the batch in slices of complexes (one slice's A1 / A2 / G2 / G3 rows alive at a time): same gradients as the
    whole batch in one launch — the dropout mask included (drop_row0), the weight-gradient partials added up across the slices' launches."""
    B, N, K = 5, 150, 32
    g = torch.Generator(device="cpu").manual_seed(5)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    leaves = [rn(B, N, K, 128), rn(B, N, 128), rn(B, N, 128), rn(128, 128, sc=0.08), rn(128, 128, sc=0.1), rn(128, sc=0.1), rn(128, 128, sc=0.1),
              rn(128, sc=0.1), 1 + rn(128, sc=0.1), rn(128, sc=0.1)]
    E32 = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(DEV).to(torch.int32).contiguous()
    R = rn(B, N, K, 128)
    monkeypatch.setattr(train, "X3", 1)
    grads = {}
    for nsl in (1, 4):
        monkeypatch.setattr(train, "EDGE_UPDATE_SLICES", nsl)
        with torch.enable_grad():
            ins = [t.clone().requires_grad_(True) for t in leaves]
            out = train._EdgeUpdate.apply(*ins, E32, p, 77)
            (out * R).sum().backward()
        grads[nsl] = [t.grad for t in ins]
    for a, b in zip(grads[1], grads[4]):
        assert rel(a, b) < 2e-6, rel(a, b)


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_edge_update_backward(p):
    """_EdgeUpdate (message + dropout3 + residual + LayerNorm3 in one launch each way).  p = 0: against fp64 autograd of the
    dense formula.  p > 0: the hash mask is regenerated by the backward launch — checked by (i) the mask statistics and the
    1/(1-p) scale recovered from a linearised forward, (ii) a directional finite difference of the deterministic (fixed
    seed) function against the analytic gradient."""
    B, N, K = 2, 41, 20
    g = torch.Generator(device="cpu").manual_seed(11)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(DEV)
    h_E, Pa, Pc = rn(B, N, K, 128), rn(B, N, 128), rn(B, N, 128)
    W1, W2, W3 = rn(128, 384, sc=0.08), rn(128, 128, sc=0.1), rn(128, 128, sc=0.1)
    b2, b3, lw, lb = rn(128, sc=0.1), rn(128, sc=0.1), 1 + rn(128, sc=0.1), rn(128, sc=0.1)
    E_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).to(DEV)
    E32 = E_idx.to(torch.int32).contiguous()
    R = rn(B, N, K, 128)
    seed = 1234
    leaves = [h_E, Pa, Pc, W1, W2, b2, W3, b3, lw, lb]
    names = ["h_E", "Pa", "Pc", "W1", "W2", "b2", "W3", "b3", "ln_w", "ln_b"]

    def ours(ts):
        hE, pa, pc, w1, w2, bb2, w3, bb3, w_, b_ = ts
        return train._EdgeUpdate.apply(hE, pa, pc, w1[:, 128:256], w2, bb2, w3, bb3, w_, b_, E32, p, seed)

    with torch.enable_grad():
        ours_in = [t.clone().requires_grad_(True) for t in leaves]
        out = ours(ours_in)
        (out * R).sum().backward()
    if p == 0.0:
        with torch.enable_grad():
            ref_in = [t.double().requires_grad_(True) for t in leaves]
            hE, pa, pc, w1, w2, bb2, w3, bb3, w_, b_ = ref_in
            bidx = torch.arange(B, device=DEV)[:, None, None]
            z1 = hE @ w1[:, 128:256].t() + pa[:, :, None] + pc[bidx, E_idx]
            z3 = F.gelu(F.gelu(z1) @ w2.t() + bb2) @ w3.t() + bb3
            ref = F.layer_norm(hE + z3, (128,), w_, b_, 1e-5)
            (ref * R.double()).sum().backward()
        assert rel(out, ref) < 2e-5
        for name, a, b in zip(names, ours_in, ref_in):
            assert rel(a.grad, b.grad) < 5e-5, (name, rel(a.grad, b.grad))
        return
    # (i) with W3 = 0 the message is b3 everywhere: LN(h_E + mask*scale*b3) exposes the mask through a probe
    with torch.no_grad():
        big = torch.full((128,), 50.0, device=DEV)
        probe = train._EdgeUpdate.apply(torch.zeros_like(h_E), Pa, Pc, W1[:, 128:256], W2, b2, torch.zeros_like(W3), big,
                                        torch.ones(128, device=DEV), torch.zeros(128, device=DEV), E32, p, seed)
        kept = probe > probe.mean(-1, keepdim=True)              # kept channels carry 50/(1-p), dropped ones 0
        frac = float(kept.float().mean())
        assert abs(frac - (1 - p)) < 0.01, frac
        again = train._EdgeUpdate.apply(torch.zeros_like(h_E), Pa, Pc, W1[:, 128:256], W2, b2, torch.zeros_like(W3), big,
                                        torch.ones(128, device=DEV), torch.zeros(128, device=DEV), E32, p, seed)
        assert torch.equal(probe, again)                         # same seed, same mask
        other = train._EdgeUpdate.apply(torch.zeros_like(h_E), Pa, Pc, W1[:, 128:256], W2, b2, torch.zeros_like(W3), big,
                                        torch.ones(128, device=DEV), torch.zeros(128, device=DEV), E32, p, seed + 1)
        assert float(((other > other.mean(-1, keepdim=True)) != kept).float().mean()) > 0.2
    # (ii) directional finite differences (fp64 accumulation of the fp32 outputs)
    eps = 1e-2
    for i, name in enumerate(names):
        v = torch.randn(leaves[i].shape, generator=g).to(DEV)
        if name == "W1":
            v[:, :128] = 0; v[:, 256:] = 0
        with torch.no_grad():
            plus = [t.clone() for t in leaves]; minus = [t.clone() for t in leaves]
            plus[i] += eps * v; minus[i] -= eps * v
            fd = float(((ours(plus).double() - ours(minus).double()) * R.double()).sum() / (2 * eps))
        an = float((ours_in[i].grad.double() * v.double()).sum())
        assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (name, fd, an)


@pytest.mark.parametrize("prec", ["x3", "bf16"])
def test_feature_weight_gradient_from_bf16_tiles(weights_np, prec, monkeypatch):
    """The backward launch of norm_edges + W_e leaves dL/dy a second time as bf16 operand tiles [tile][channel][64 edges] (hi | remainder); the
    embedding-weight gradient stages those by 16-byte copies instead of converting / transposing the fp32 rows.  Same operands, same products:
    the gradients are bit-identical to the fp32-row path (here: the tiles withheld), at a size with a partial last tile (46 x 14 = 644 edges)."""
    n, k = 46, 14
    cx = synth.make_complex(seed=31, n=n, n_chains=2, missing_atom_frac=0.05, masked_frac=0.04)
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(DEV) for k_, v in cx.items()}
    m = make_model(weights_np, k)
    m.message_precision = prec
    monkeypatch.setattr(train, "X3", train.PREC_CODE[prec])
    monkeypatch.setattr(train, "G16_SPLIT", True)             # (split-bf16: the hi + remainder tiles are off by default)
    fp = m.features
    R = None
    grads = []

    class NoTiles(dict):
        def __setitem__(self, key, value):
            pass

    for tiles in (True, False):
        if not tiles:
            monkeypatch.setattr(train, "_G16", NoTiles())
        m.zero_grad()
        with torch.enable_grad():
            y, E_idx = train.edge_embedding(m, fd)
            h = train._EdgeEmbedTail.apply(y, fp.norm_edges.weight, fp.norm_edges.bias, m.W_e.weight, m.W_e.bias)
            if R is None:
                R = torch.randn(h.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
            (h * R).sum().backward()
        grads.append({n_: p.grad.clone() for n_, p in fp.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and "edge_embedding.weight" in grads[0]
    for n_ in grads[0]:
        assert torch.equal(grads[0][n_], grads[1][n_]), n_


def test_feature_weight_gradient(weights_np):
    """namp_train_feat_wgrad (+ the positional path) vs torch autograd through the dense RBF featurisation."""
    n, k = 46, 14
    cx = synth.make_complex(seed=31, n=n, n_chains=2, missing_atom_frac=0.05, masked_frac=0.04)
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(DEV) for k_, v in cx.items()}
    m = make_model(weights_np, k)
    fp = m.features
    with torch.enable_grad():
        y, E_idx = train.edge_embedding(m, fd)
        R = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
        (y * R).sum().backward()
    got = {n_: p.grad.clone() for n_, p in fp.named_parameters() if p.grad is not None}
    m.zero_grad()
    # dense torch formulation of the same features on the same neighbour lists
    X18, M18 = train._atom_frames(m, fd["X"].float(), fd)
    j = E_idx.long()
    bidx = torch.arange(1, device=DEV)[:, None, None]
    D = torch.sqrt(((X18[:, :, None, :, None, :] - X18[bidx, j][:, :, :, None, :, :]) ** 2).sum(-1) + 1e-6)
    mu = torch.linspace(2., 22., 16, device=DEV)
    rbf = torch.exp(-(((D[..., None] - mu) / 1.25) ** 2)) * M18[:, :, None, :, None, None] * M18[bidx, j][:, :, :, None, :, None]
    R_idx, ch = fd["R_idx"].long(), fd["chain_labels"].long()
    same = (ch[:, :, None] == ch[bidx, j]).long()
    d = torch.clip(R_idx[:, :, None] - R_idx[bidx, j] + 32, 0, 64) * same + (1 - same) * 65
    with torch.enable_grad():
        pos = fp.embeddings.linear.weight.t()[d] + fp.embeddings.linear.bias
        feat = torch.cat((pos, rbf.reshape(1, n, k, -1)), -1)
        y_ref = feat @ fp.edge_embedding.weight.t()
        (y_ref * R).sum().backward()
    assert rel(y, y_ref) < 1e-4
    for n_, p in fp.named_parameters():
        if p.grad is None:
            continue
        assert rel(got[n_], p.grad) < 1e-4, (n_, rel(got[n_], p.grad))
    assert {"edge_embedding.weight", "embeddings.linear.weight", "embeddings.linear.bias"} <= set(got)


@pytest.mark.parametrize("rows", [1, 7, 64, 1000, 70001])
def test_row_layernorm_matches_torch(rows):
    """ln_rows_fwd / ln_rows_bwd (norm_edges on the edge embedding) against torch.nn.functional.layer_norm in fp64."""
    torch.manual_seed(rows)
    x = (torch.randn(rows, 128, device=DEV) * 3 + 0.5).requires_grad_()
    w = (torch.randn(128, device=DEV) * 0.5 + 1).requires_grad_()
    b = torch.randn(128, device=DEV).requires_grad_()
    gout = torch.randn(rows, 128, device=DEV)
    with torch.enable_grad():
        y = train._RowLayerNorm.apply(x, w, b)
        y.backward(gout)
    xr, wr, br = (t.detach().double().requires_grad_() for t in (x, w, b))
    with torch.enable_grad():
        yr = torch.nn.functional.layer_norm(xr, (128,), wr, br, 1e-5)
        yr.backward(gout.double())
    assert rel(y, yr) < 2e-6
    assert rel(x.grad, xr.grad) < 2e-5 and rel(w.grad, wr.grad) < 2e-5 and rel(b.grad, br.grad) < 2e-5


def _g7_on_device():
    fd, k = g7_inputs()
    return {k_: v.to(DEV) for k_, v in fd.items()}, k


@pytest.mark.parametrize("tag", ["g7_training", "g7b_training"])
@pytest.mark.parametrize("prec", ["x3", "fp32"])
def test_training_step_matches_reference_golden(golden_dir, weights_np, prec, tag):
    """G7: loss, gradients of all 123 parameters and the first Noam/Adam update of na_run.py:198-238 — with the per-edge
    GEMMs of forward, backward and weight gradients as split-bf16 products (default) and as exact fp32 MFMA.
    G7b: the same step with a non-empty ppm_mask / aligned_ppm (the specificity model's target, na_model_utils.py:134)."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    fd, k = _g7_on_device()
    if tag == "g7b_training":
        pm, ppm = g7b_ppm(g7_inputs()[0])
        assert int(pm.sum()) >= 8
        fd["ppm_mask"], fd["aligned_ppm"] = pm.to(DEV), ppm.to(DEV)
    m = make_model(weights_np, k).train()
    m.message_precision = prec
    opt = train.get_std_opt(m.parameters(), 128, 0)
    rti = spec.restype_to_int()
    rm, rn = train.polymer_restype_tables(rti, 33, DEV)
    no_loss = torch.tensor([rti[t] for t in cpu_ref.NO_LOSS_TOKENS], device=DEV)
    with torch.enable_grad():
        loss, lp = train.train_step(m, opt, fd, rm, rn, no_loss, decoding_randn=torch.from_numpy(g["randn"]).to(DEV))
    assert float((lp.cpu() - torch.from_numpy(g["log_probs"])).abs().max()) < 1e-3
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert abs(opt._rate - float(g["lr_step1"])) < 1e-15
    names = [str(n) for n in g["names"]]
    params = dict(m.named_parameters())
    assert sorted(params) == names
    worst = 0.0
    for i, n in enumerate(names):
        gr = params[n].grad
        assert gr is not None, n
        scale = float(g["grad_absmax"][i]) + 1e-12
        pick = gr.reshape(-1)[torch.from_numpy(g["pick"][i]).to(DEV)].cpu().numpy()
        err = float(np.abs(pick - g["grad_pick"][i]).max()) / scale
        nerr = abs(float(gr.double().norm()) - float(g["grad_norm"][i])) / (float(g["grad_norm"][i]) + 1e-12)
        worst = max(worst, err, nerr)
        assert err < 1e-4 and nerr < 1e-4, (n, err, nerr)       # measured: 1.8e-6
    print(f"worst relative gradient error vs reference: {worst:.2e}")
    # the first Noam / Adam update (multi-tensor HIP step, train.FusedAdam) against the reference's NoamOpt + torch.optim.Adam step:
    # at step 1 every entry moves by -lr * g / (|g| + eps), i.e. ~lr in magnitude; compared to 2 % of lr (a parameter's fp32 ulp is
    # ~1 % of lr, and entries whose gradient is within 1e-9 of zero are ambiguous)
    w0 = {n: torch.from_numpy(weights_np[n]) for n in names}
    lr1 = float(g["lr_step1"])
    for i, n in enumerate(names):
        pk = torch.from_numpy(g["pick"][i])
        delta = (params[n].detach().cpu() - w0[n]).reshape(-1)[pk].numpy()
        big = np.abs(g["grad_pick"][i]) > 1e-7
        assert np.abs(delta - g["delta_pick"][i])[big].max(initial=0.0) < 0.02 * lr1 + 1.5e-8, n


@pytest.mark.parametrize("ns,k", [([9], 48), ([20, 7], 5), ([40, 33, 24], 17), ([65], 50), ([130, 90], 33)])
def test_training_gradients_odd_shapes(weights_np, ns, k):
    """Loss and all parameter gradients against the oracle's autograd for padded batches with K from 5 to 64, K > L, and
    residues without any earlier neighbour.  (Every complex keeps more unmasked residues than K, or runs alone with K >= L:
    otherwise the farthest real residue ties with the masked / padded ones at the row maximum and torch.topk's choice among
    them — the reference's neighbour list — is unspecified.)"""
    cxs = [synth.make_complex(seed=600 + 10 * i + n, n=n, n_chains=min(3, n), masked_frac=0.1 if n > 15 else 0.0) for i, n in enumerate(ns)]
    from na_mpnn_amd import shard
    fd = shard.pad_batch(cxs)
    fd["S"] = fd["S"].long()
    Lmax = max(ns)
    randn = torch.randn(len(ns), Lmax, generator=torch.Generator().manual_seed(k))
    rti = spec.restype_to_int()
    loss_ref, _, g_ref = cpu_ref.train_loss_and_grads({k_: torch.from_numpy(v) for k_, v in weights_np.items()}, fd, k, randn, rti)
    m = make_model(weights_np, k).train()
    fdd = {k_: v.to(DEV) for k_, v in fd.items()}
    rm, rn = train.polymer_restype_tables(rti, 33, DEV)
    no_loss = torch.tensor([rti[t] for t in cpu_ref.NO_LOSS_TOKENS], device=DEV)
    opt = train.get_std_opt(m.parameters(), 128, 0)
    with torch.enable_grad():
        loss, _ = train.train_step(m, opt, fdd, rm, rn, no_loss, decoding_randn=randn.to(DEV))
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1e-3, abs(float(loss_ref)))
    for name, p in m.named_parameters():
        gr, ref = p.grad.detach().cpu().double(), g_ref[name].double()
        scale = float(ref.abs().max())
        if scale < 1e-12:
            assert float(gr.abs().max()) < 1e-9, name
        else:
            assert float((gr - ref).abs().max()) / scale < 2e-4, (name, float((gr - ref).abs().max()) / scale)


def test_training_gradients_without_pred_na_N():
    """include_pred_na_N = 0: loss and every gradient — in particular of the [128 x 4640] edge embedding, which reaches the
    kernels through the zero-padded 18-atom expansion — against the oracle's autograd."""
    w = synth.make_weights_noN(0)
    k = 20
    cxs = [synth.make_complex(seed=640 + n, n=n, n_chains=2, masked_frac=0.05) for n in (44, 31)]
    from na_mpnn_amd import shard
    fd = shard.pad_batch(cxs)
    fd["S"] = fd["S"].long()
    randn = torch.randn(2, 44, generator=torch.Generator().manual_seed(3))
    rti = spec.restype_to_int()
    loss_ref, _, g_ref = cpu_ref.train_loss_and_grads({k_: torch.from_numpy(v) for k_, v in w.items()}, fd, k, randn, rti)
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, dropout=0.0, atom_dict=spec.atom_dict(), restype_to_int=rti,
                    polytype_to_int=spec.polytype_to_int(), include_pred_na_N=0)
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()})
    m = m.to(DEV).train()
    fdd = {k_: v.to(DEV) for k_, v in fd.items()}
    rm, rn = train.polymer_restype_tables(rti, 33, DEV)
    no_loss = torch.tensor([rti[t] for t in cpu_ref.NO_LOSS_TOKENS], device=DEV)
    opt = train.get_std_opt(m.parameters(), 128, 0)
    with torch.enable_grad():
        loss, _ = train.train_step(m, opt, fdd, rm, rn, no_loss, decoding_randn=randn.to(DEV))
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1e-3, abs(float(loss_ref)))
    for name, p in m.named_parameters():
        gr, ref = p.grad.detach().cpu().double(), g_ref[name].double()
        assert gr.shape == ref.shape, name
        scale = float(ref.abs().max())
        if scale < 1e-12:
            assert float(gr.abs().max()) < 1e-9, name
        else:
            assert float((gr - ref).abs().max()) / scale < 2e-4, (name, float((gr - ref).abs().max()) / scale)


def test_training_reduces_loss_with_dropout(weights_np):
    """Twelve steps on a fixed batch in train mode with dropout 0.1 and coordinate noise: finite grads, loss goes down;
    the updated weights still drive the inference kernels (repack on parameter version change)."""
    fd, k = _g7_on_device()
    m = make_model(weights_np, k, dropout=0.1).train()
    m.protein_augment_eps = m.dna_augment_eps = m.rna_augment_eps = 0.02
    rti = spec.restype_to_int()
    rm, rn = train.polymer_restype_tables(rti, 33, DEV)
    no_loss = torch.tensor([rti[t] for t in cpu_ref.NO_LOSS_TOKENS], device=DEV)
    opt = train.NoamOpt(128, 1, 100, torch.optim.Adam(m.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9), 0)
    torch.manual_seed(0)
    fixed = torch.randn(2, 72, device=DEV)
    S = fd["S"].long()
    mfl = fd["mask"] * (1 - torch.any(S[:, :, None] == no_loss[None, None, :], dim=-1).long())
    pm = {"protein": fd["protein_mask"], "dna": fd["dna_mask"], "rna": fd["rna_mask"]}

    def eval_loss():
        m.eval()
        with torch.no_grad():
            lp, _ = m(fd, decoding_randn=fixed)
        m.train()
        return float(train.loss_smoothed(S, lp, mfl, pm, rm, rn)[1])

    before = eval_loss()
    with torch.enable_grad():
        for _ in range(12):
            loss, _ = train.train_step(m, opt, fd, rm, rn, no_loss, gradient_norm=1.0)
            assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    after = eval_loss()
    assert after < 0.97 * before, (before, after)
    m.eval()
    with torch.no_grad():
        fd1 = dict(fd); fd1["batch_size"] = 1; fd1["chain_mask"] = fd["mask"]; fd1["randn"] = torch.randn(2, 72, device=DEV)
        lp = m.score(fd1)["log_probs"]
        w = {k_: v.detach().cpu() for k_, v in m.state_dict().items()}
        ref = cpu_ref.score(w, {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd1.items()}, k)["log_probs"]
    valid = fd["mask"].bool().cpu()
    assert float((lp.cpu() - ref)[valid].abs().max()) < 1e-3


def test_checkpoint_round_trip(tmp_path, weights_np):
    fd, k = _g7_on_device()
    m = make_model(weights_np, k)
    opt = train.get_std_opt(m.parameters(), 128, 7)
    path = str(tmp_path / "ck.pt")
    train.save_checkpoint(path, m, opt, epoch=3, step=7)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "step", "save_step", "model_state_dict", "optimizer_state_dict"}
    assert sorted(ck["model_state_dict"]) == sorted(weights_np)
    m2 = make_model(synth.make_weights(5), k)
    train.load_checkpoint(path, m2, map_location="cpu")
    for (n1, p1), (n2, p2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2.cpu())


def test_cfg5_sized_training_step(weights_np):
    """BASELINE configs[4] at its real size (B=16 x N=1500, K=48: 1.15 M edges, ~11 GiB of transients): one optimisation
    step in the split-bf16 and in the exact-fp32 evaluation — finite loss and gradients, and the two losses agree to 1e-4 —, and in the
    mixed-precision (bf16) mode the reference trains in (na_run.py:216-238): its loss within 0.5 % of the fp32 step's and the whole gradient
    vector within 2 % of the fp32 step's in direction (cosine > 0.999)."""
    B, N, K = 16, 1500, 48
    rti = spec.restype_to_int()
    cxs = [synth.make_complex(seed=5000 + b, n=N, n_chains=4) for b in range(B)]
    fd = {k_: torch.from_numpy(np.stack([c[k_] for c in cxs])).to(DEV) for k_ in cxs[0]}
    fd["S"] = fd["S"].long()
    randn = torch.randn(B, N, generator=torch.Generator().manual_seed(5)).to(DEV)
    rm, rn = train.polymer_restype_tables(rti, 33, DEV)
    no_loss = torch.tensor([rti[t] for t in cpu_ref.NO_LOSS_TOKENS], device=DEV)
    losses, gvec = {}, {}
    for prec in ("x3", "fp32", "bf16"):
        m = make_model(weights_np, K).train()
        m.message_precision = prec
        opt = train.get_std_opt(m.parameters(), 128, 0)
        with torch.enable_grad():
            loss, lp = train.train_step(m, opt, fd, rm, rn, no_loss, loss_tokens=6000.0, gradient_norm=1.0, decoding_randn=randn)
        assert torch.isfinite(loss) and torch.isfinite(lp).all()
        for name, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
        losses[prec] = float(loss)
        gvec[prec] = torch.cat([p.grad.detach().double().flatten() for _, p in sorted(m.named_parameters())]).cpu()
        del m, opt
        torch.cuda.empty_cache()
    assert abs(losses["x3"] - losses["fp32"]) <= 1e-4 * abs(losses["fp32"]), losses
    assert abs(losses["bf16"] - losses["fp32"]) <= 5e-3 * abs(losses["fp32"]), losses
    cos = float((gvec["bf16"] @ gvec["fp32"]) / (gvec["bf16"].norm() * gvec["fp32"].norm()))
    print(f"cfg5-sized step: losses {losses}, cos(grad bf16, grad fp32) = {cos:.6f}")
    assert cos > 0.999, cos


def test_mixed_precision_training_mode(weights_np):
    """The reference trains under torch.cuda.amp.autocast + GradScaler (na_run.py:21,216-238).  Here: message_precision
    "bf16" = per-edge GEMMs of forward / backward / weight gradients as plain bf16 products (fp32 accumulate, fp32 master
    weights).  PARITY: the step against the oracle's CPU emulation of exactly those rounding points (oracle/cpu_ref_mixed.py:
    bf16-rounded operands of every per-edge product, forward, data gradient and weight gradient; fp32 elsewhere) — loss within
    0.2 %, every gradient within 2 % of the emulation's (relative to its norm; the fp32 step itself sits ~0.5 % away from both).
    Then: torch's GradScaler usable unchanged, and a few steps reduce the loss."""
    from oracle import cpu_ref_mixed
    fd, k = _g7_on_device()
    fd_cpu = g7_inputs()[0]
    rti = spec.restype_to_int()
    rm, rn = train.polymer_restype_tables(rti, 33, DEV)
    no_loss = torch.tensor([rti[t] for t in cpu_ref.NO_LOSS_TOKENS], device=DEV)
    randn_cpu = torch.randn(fd["mask"].shape, generator=torch.Generator().manual_seed(11))
    randn = randn_cpu.to(DEV)
    grads, losses = {}, {}
    for prec in ("fp32", "bf16"):
        m = make_model(weights_np, k).train()
        m.message_precision = prec
        opt = train.get_std_opt(m.parameters(), 128, 0)
        with torch.enable_grad():
            loss, _ = train.train_step(m, opt, fd, rm, rn, no_loss, decoding_randn=randn)
        losses[prec] = float(loss)
        grads[prec] = {n: p.grad.detach().double().flatten().cpu() for n, p in m.named_parameters()}
    wt = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    l_em, _, g_em = cpu_ref_mixed.train_loss_and_grads(wt, fd_cpu, k, randn_cpu, rti)
    assert abs(losses["bf16"] - float(l_em)) < 2e-3 * abs(float(l_em)), (losses, float(l_em))
    worst = ("", 0.0)
    gnorm = float(torch.cat([g_em[n].double().flatten() for n in g_em]).norm())
    for n in grads["bf16"]:
        gb, ge = grads["bf16"][n], g_em[n].double().flatten()
        if float(ge.norm()) > 1e-6 * gnorm:
            r = float((gb - ge).norm() / ge.norm())
            if r > worst[1]:
                worst = (n, r)
            assert r < 0.02, (n, r)
    d32 = max(float((grads["fp32"][n] - g_em[n].double().flatten()).norm() / g_em[n].double().norm()) for n in g_em
              if float(g_em[n].double().norm()) > 1e-6 * gnorm)
    print(f"mixed precision vs CPU emulation: loss {losses['bf16']:.6f} / {float(l_em):.6f}; worst gradient {worst[0]} {worst[1]:.4f} "
          f"(the fp32 step is {d32:.4f} away from the emulation)")
    a = torch.cat([grads["bf16"][n] for n in sorted(grads["fp32"])])
    b = torch.cat([grads["fp32"][n] for n in sorted(grads["fp32"])])
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.999, cos
    # GradScaler around the same step (the reference's call sequence), a few steps: finite and decreasing
    m = make_model(weights_np, k).train()
    m.message_precision = "bf16"
    opt = train.NoamOpt(128, 1, 100, torch.optim.Adam(m.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9), 0)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    hist = []
    with torch.enable_grad():
        for _ in range(10):
            loss, _ = train.train_step(m, opt, fd, rm, rn, no_loss, decoding_randn=randn, gradient_norm=0.0, scaler=scaler)
            hist.append(float(loss))
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("G", [37, 64, 300])
def test_node_tail_matches_autograd(G, p, monkeypatch):
    """_NodeTail (residue tail of EncLayer / DecLayer in training: LayerNorm1, FFN, LayerNorm2, mask, both dropouts — one HIP
    launch each way) against fp64 autograd of the same expression.  With p > 0 the kernels' hash dropout masks are read back
    through two probe calls (they depend on seed, row and channel only) and handed to the reference."""
    monkeypatch.setattr(train, "X3", 1)                      # split-bf16 products, whatever an earlier test's forward_train left
    gen = torch.Generator(device="cpu").manual_seed(1000 + G)
    rn = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=gen)).to(DEV)
    h_V, dh, R = rn(G, 128), rn(G, 128, sc=0.5), rn(G, 128)
    mask = (torch.rand(G, generator=gen) > 0.2).to(DEV)
    ln1_w, ln1_b, ln2_w, ln2_b = 1 + rn(128, sc=0.1), rn(128, sc=0.1), 1 + rn(128, sc=0.1), rn(128, sc=0.1)
    W_in, b_in, W_out, b_out = rn(512, 128, sc=0.08), rn(512, sc=0.1), rn(128, 512, sc=0.05), rn(128, sc=0.1)
    mask32 = mask.to(torch.int32).contiguous()
    s1, s2 = 12345, 67890
    ones, zeros = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    if p > 0:
        scale = 1.0 / (1.0 - p)
        # probe through the ABI: h_V = 0, dh = 1 -> LayerNorm1's input is the mask itself (kept entries sit above the row
        # mean); W_out = 0, b_out = 1 -> y - x1 is the second mask
        from na_mpnn_amd import hip as H_
        L = H_.lib()
        x1 = torch.empty(G, 128, device=DEV); y = torch.empty_like(x1); out = torch.empty_like(x1); z = torch.empty(4, G, 128, device=DEV)
        # every operand is a NAMED tensor: an inline temporary would be freed (and its block re-used by the next temporary's
        # fill) before the launch that reads it is even enqueued
        zi, zo = train._ximage_general(torch.zeros_like(W_in)), train._ximage_general(torch.zeros_like(W_out))
        hv0, dh1, bin0 = torch.zeros(G, 128, device=DEV), torch.ones(G, 128, device=DEV), torch.zeros(512, device=DEV)
        H_.check(L.namp_train_tail_fwd(hv0.data_ptr(), dh1.data_ptr(), None, ones.data_ptr(), zeros.data_ptr(), zi.data_ptr(),
                                       bin0.data_ptr(), zo.data_ptr(), ones.data_ptr(), ones.data_ptr(), zeros.data_ptr(), float(p), s1, s2,
                                       out.data_ptr(), x1.data_ptr(), z.data_ptr(), y.data_ptr(), G, H_.current_stream()))
        torch.cuda.synchronize()
        m1 = (x1 > 0).double() * scale                       # dropout1 mask (kept entries lie above the row mean)
        m2 = (y - x1).double()                               # dropout2 mask * scale (f == b_out == 1 everywhere)
        assert 0.6 < float((m1 > 0).double().mean()) < 0.95 and 0.6 < float((m2 > 0).double().mean()) < 0.95
        assert float((m2[m2 > 0] - scale).abs().max()) < 1e-5
    else:
        m1 = m2 = torch.ones(G, 128, device=DEV, dtype=torch.float64)
    leaves = [h_V, dh, ln1_w, ln1_b, W_in, b_in, W_out, b_out, ln2_w, ln2_b]

    def dense(hV, d, l1w, l1b, Wi, bi, Wo, bo, l2w, l2b):
        x1_ = F.layer_norm(hV + m1 * d, (128,), l1w, l1b, 1e-5)
        f = F.linear(F.gelu(F.linear(x1_, Wi, bi)), Wo, bo)
        return mask.double().unsqueeze(-1) * F.layer_norm(x1_ + m2 * f, (128,), l2w, l2b, 1e-5)

    with torch.enable_grad():
        ref_in = [t.double().requires_grad_(True) for t in leaves]
        ref_out = dense(*ref_in)
        (ref_out * R.double()).sum().backward()
        ours_in = [t.clone().requires_grad_(True) for t in leaves]
        hV_, d_, l1w_, l1b_, Wi_, bi_, Wo_, bo_, l2w_, l2b_ = ours_in
        out = train._NodeTail.apply(hV_, d_, mask32, l1w_, l1b_, Wi_, bi_, Wo_, bo_, l2w_, l2b_, p, s1, s2)
        (out * R).sum().backward()
    assert rel(out, ref_out) < 2e-5, rel(out, ref_out)
    names = ["h_V", "dh", "ln1_w", "ln1_b", "W_in", "b_in", "W_out", "b_out", "ln2_w", "ln2_b"]
    for name, a_, b_ in zip(names, ours_in, ref_in):
        assert a_.grad is not None, name
        assert rel(a_.grad, b_.grad) < 1e-4, (name, rel(a_.grad, b_.grad))


@pytest.mark.parametrize("ppm", [False, True])
def test_fused_loss_matches_the_stock_op_form(ppm):
    """train.loss_smoothed on the device (one HIP launch each way, fp64) against its stock-op restatement of na_model_utils.py:111-146
    evaluated on the host with the same inputs: per-residue loss, reduced loss and the gradient with respect to log_probs."""
    g = torch.Generator().manual_seed(17 + int(ppm))
    B, L, V = 3, 70, 33
    rti = spec.restype_to_int()
    rm_d, rn = train.polymer_restype_tables(rti, V, DEV)
    rm_c = {k_: v.cpu() for k_, v in rm_d.items()}
    lp = torch.log_softmax(torch.randn(B, L, V, generator=g), -1)
    S = torch.randint(0, V, (B, L), generator=g)
    poly = torch.randint(0, 4, (B, L), generator=g)                       # 3 = none of the polymers (masked residue)
    pmask = {"protein": (poly == 0).int(), "dna": (poly == 1).int(), "rna": (poly == 2).int()}
    mask = (torch.rand(B, L, generator=g) > 0.1).int()
    kw = {}
    if ppm:
        kw = {"ppm_mask": (torch.rand(B, L, generator=g) > 0.6).int(), "aligned_ppm": torch.softmax(torch.randn(B, L, V, generator=g), -1).double()}
    with torch.enable_grad():
        lp_c = lp.clone().requires_grad_(True)
        per_c, red_c = train.loss_smoothed(S, lp_c, mask, pmask, rm_c, rn, weight=0.1, tokens=6000.0, num_letters=V, **kw)
        red_c.backward()
        lp_d = lp.to(DEV).requires_grad_(True)
        kw_d = {k_: v.to(DEV) for k_, v in kw.items()}
        per_d, red_d = train.loss_smoothed(S.to(DEV), lp_d, mask.to(DEV), {k_: v.to(DEV) for k_, v in pmask.items()}, rm_d, rn, weight=0.1,
                                           tokens=6000.0, num_letters=V, **kw_d)
        red_d.backward()
    assert per_d.dtype == torch.float64
    assert float((per_d.cpu() - per_c.detach()).abs().max()) < 1e-12
    assert abs(float(red_d) - float(red_c)) < 1e-13
    assert float((lp_d.grad.cpu() - lp_c.grad).abs().max()) < 1e-10


@pytest.mark.parametrize("clip", [0.0, 0.5])
def test_fused_adam_matches_torch_adam(clip):
    """train.FusedAdam (one multi-tensor launch; with clipping: norm partials + coefficient + step) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam over five steps on tensors of awkward sizes; its state_dict loads into a plain
    torch.optim.Adam (the reference's checkpoint format, na_run.py:342)."""
    g = torch.Generator().manual_seed(5)
    shapes = [(128, 384), (128,), (33, 128), (1,), (512, 128), (2049,), (128, 5200)]
    pa = [torch.nn.Parameter(torch.randn(*sh, generator=g).to(DEV)) for sh in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = train.NoamOpt(128, 2, 4000, train.FusedAdam(pa, lr=0, betas=(0.9, 0.98), eps=1e-9), 0)
    ob = train.NoamOpt(128, 2, 4000, torch.optim.Adam(pb, lr=0, betas=(0.9, 0.98), eps=1e-9), 0)
    for it in range(5):
        grads = [torch.randn(*sh, generator=g).to(DEV) * (10.0 ** (it - 2)) for sh in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.optimizer.clip_norm = clip
        if clip > 0:
            ref_norm = torch.nn.utils.clip_grad_norm_(pb, clip)
        oa.step(); ob.step()
        if clip > 0:
            assert abs(float(oa.optimizer.last_grad_norm[0]) - float(ref_norm)) < 1e-5 * float(ref_norm)
            for p, q in zip(pa, pb):
                assert float((p.grad - q.grad).abs().max()) <= 1e-6 * float(q.grad.abs().max())
        for p, q in zip(pa, pb):
            # an Adam update is ~lr per entry whatever the gradient scale: compare the parameters to a fraction of the step
            assert float((p - q).abs().max()) < 0.02 * oa._rate + 1e-7 * float(q.abs().max()), (it, tuple(p.shape))
            sa, sb = oa.optimizer.state[p], ob.optimizer.state[q]
            assert float(sa["step"]) == float(sb["step"]) == it + 1
            assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-6 * float(sb["exp_avg"].abs().max()) + 1e-12
            assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-6 * float(sb["exp_avg_sq"].abs().max()) + 1e-20
    plain = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in pa], lr=0, betas=(0.9, 0.98), eps=1e-9)
    plain.load_state_dict(oa.optimizer.state_dict())
    assert float(plain.state[plain.param_groups[0]["params"][0]]["step"]) == 5.0

