"""Training path: differentiable forward of ``ProteinMPNN`` on the HIP kernels, the
label-smoothed loss, the Noam/Adam schedule and the checkpoint file format of the reference's ``na_run.py``.

Split of work
    per-edge ([B,N,K,128]-sized) forward and backward   -> libnamp_hip.so (``namp_train_*``; csrc/namp_train.h)
    per-residue ([B,N,128]-sized) ops, logits, loss     -> torch autograd on the device (LayerNorm, FFN, the hoisted
                                                           first-layer tables, W_out, log_softmax: < 2 % of the flops)
The reference checkpoints every layer (``torch.utils.checkpoint``, na_model_utils.py:606,637): only layer inputs
survive the forward pass and activations are recomputed in backward.  ``_EdgeMLP`` keeps the same policy — it
saves its inputs and ``namp_train_edge_bwd`` recomputes the chain in registers.

There is no CPU fallback: every tensor must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

import torch
import torch.nn.functional as F

from . import hip, spec

H = 128
ENC_MSG, DEC_MSG, ENC_EDGE = 0, 1, 2


# Evaluation of the per-edge GEMMs of the training kernels (forward, backward, weight gradients) — the `x3` argument of the
# namp_train_* entry points: 1 = split-bf16 products (fp32-equivalent to ~2^-16, the default, like the inference path's
# "x3" mode); 0 = exact fp32 MFMA; 2 = plain bf16 products with fp32 accumulation — the MIXED-PRECISION mode, mirroring the
# reference's training under torch.cuda.amp.autocast (na_run.py:21,216-238; master weights, residue-level math, losses
# and the optimiser stay fp32, and torch's GradScaler can be used unchanged around loss.backward()).  Set through
# forward_train from model.message_precision ("fp32" / "x3" / "bf16").
X3 = 1
PREC_CODE = {"fp32": 0, "x3": 1, "bf16": 2}
# Message-stage backward that accumulates its weight gradients on chip (csrc/namp_train_dw.h; split-bf16 and bf16 products).
# NAMP_TRAIN_DW=0 restores the row tensors + row-contraction launches of rounds 1-3 (A/B runs, tests of both forms).
import itertools
import os as _os
DW_ONCHIP = _os.environ.get("NAMP_TRAIN_DW", "1") != "0"
# The edge update's backward in the same form (mixed precision): two persistent launches cut at g2 = dL/dz2, all three weight gradients and the
# LayerNorm sums on chip (csrc/namp_train_eu.h; round 5).  NAMP_TRAIN_DW_EDGE=0 restores the round-3 launch + its three row contractions.
DW_ONCHIP_EDGE = _os.environ.get("NAMP_TRAIN_DW_EDGE", "1") != "0"
# norm_edges + W_e as one forward / one backward launch without the normalised rows in memory (split-bf16 / mixed precision; 0 = the round-4 pair)
EMBED_LN_FUSED = _os.environ.get("NAMP_TRAIN_EMBED_LN", "1") != "0"
G16_SPLIT = _os.environ.get("NAMP_TRAIN_G16_SPLIT", "0") == "1"      # bf16 operand tiles of dL/dy in the split-bf16 mode too (measured: a loss)
# split-bf16 edge-update backward (fp32 row tensors): complexes per batch are walked in this many slices at most (1 = the whole batch at once)
EDGE_UPDATE_SLICES = max(1, int(_os.environ.get("NAMP_TRAIN_EU_SLICES", "2")))


# Fragment images made during ONE training step (forward_train and the backward pass that follows it), keyed by (storage
# address, strides, precision, transposed): a stage's forward images (W1b, W2, W3) are needed again by its backward launch,
# which recomputes the chain — pack them once.  The cache belongs to the step token forward_train creates: a new step starts
# empty, and calls outside forward_train (tests driving one autograd Function) never touch it.
_IMG_CACHE = {}
_STEP = None                      # token of the step being recorded by forward_train (None outside)
_CACHE_STEP = None                # token the cache's entries belong to


class _PackPlan:
    """All fragment images of a training step from ONE launch (namp_pack_images).  Learned, not declared: `_image` / `_ximage_general` register
    every (storage, strides, kind, transposed) they are asked for; from the next step on forward_train packs all registered blocks into one
    arena with a single launch and the same calls return arena slices.  Blocks nobody asked for during a whole step are forgotten (a replaced
    parameter set); registered blocks are kept alive by the plan, so a stale address is never read."""

    KIND_BYTES = {1: lambda o, i: 4 * o * i, 2: lambda o, i: 2 * o * i, 3: lambda o, i: 4 * o * i}

    def __init__(self):
        self.entries = {}            # key -> dict(src, kind, transposed, out_f, in_f, used, img)
        self.dirty = True
        self.arena = self.table = None
        self.nblocks = 0
        self.valid_step = None

    @staticmethod
    def key(block, kind, transposed):
        return (block.data_ptr(), tuple(block.shape), block.stride(0), block.stride(1), kind, bool(transposed), block.device.index)

    def register(self, block, kind, transposed):
        o, i = (block.shape[1], block.shape[0]) if transposed else (block.shape[0], block.shape[1])
        self.entries[self.key(block, kind, transposed)] = dict(src=block, kind=kind, transposed=bool(transposed), out_f=o, in_f=i, used=True, img=None)
        self.dirty = True

    def lookup(self, block, kind, transposed, step):
        if self.valid_step is not step:
            return None
        e = self.entries.get(self.key(block, kind, transposed))
        if e is None or e["img"] is None:
            return None
        e["used"] = True
        if e.get("version") != block._version:            # edited in place since begin_step packed it (weight tying, clamping, an EMA swap
            return None                                   # inside the step): the caller packs this block on its own
        return e["img"]

    def begin_step(self, step, dev):
        stale = [k for k, e in self.entries.items() if not e["used"] or k[-1] != dev.index]
        for k in stale:
            del self.entries[k]
        self.dirty = self.dirty or bool(stale)
        self.valid_step = None
        if not self.entries:
            return
        if self.dirty:
            off, blocks, rows = 0, 0, []
            for e in self.entries.values():
                nbytes = self.KIND_BYTES[e["kind"]](e["out_f"], e["in_f"])
                e["off"], e["nbytes"], e["first"] = off, nbytes, blocks
                off += (nbytes + 255) // 256 * 256
                blocks += (e["out_f"] * e["in_f"] + 255) // 256
            self.arena = torch.empty(off // 4, dtype=torch.float32, device=dev)
            base = self.arena.data_ptr()
            tab = (hip.NampPack * len(self.entries))()
            for q, e in enumerate(self.entries.values()):
                src = e["src"]
                assert src.stride(1) == 1 and src.dtype == torch.float32
                tab[q] = hip.NampPack(src.data_ptr(), base + e["off"], src.stride(0), e["out_f"], e["in_f"], e["kind"], int(e["transposed"]), e["first"])
                e["img"] = self.arena[e["off"] // 4:(e["off"] + e["nbytes"]) // 4]
            raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
            self.table = raw.to(dev)
            self.nblocks, self.dirty = blocks, False
        for e in self.entries.values():
            e["used"] = False
            e["version"] = e["src"]._version
        hip.check(hip.lib().namp_pack_images(self.table.data_ptr(), len(self.entries), self.nblocks, hip.current_stream()), "pack_images")
        self.valid_step = step


_PLAN = _PackPlan()


def _image(block, x3=None, transposed=False, step=None, plan=True):
    """Fragment image (fp32, or x3 / bf16) of a [128 x 128] block given as a (possibly column-sliced) view of an nn.Linear
    weight (or of its transpose).  x3 None: the module-level setting (forward passes); backward passes hand in the value
    their forward saved.  Inside a training step (split-bf16 / bf16) the image comes out of the step's one packing launch (_PackPlan)."""
    prec = int(X3 if x3 is None else x3)
    step = _STEP if step is None else step
    use_cache = step is not None and step is _CACHE_STEP
    if use_cache and prec in (1, 2):
        hit = _PLAN.lookup(block, prec, transposed, step)
        if hit is not None:
            return hit
    key = (block.data_ptr(), block.stride(0), block.stride(1), block._version, prec, transposed, block.device.index)
    if use_cache:
        hit = _IMG_CACHE.get(key)
        if hit is not None:
            return hit
        if plan and prec in (1, 2) and block.stride(1) == 1 and block.dtype == torch.float32 and tuple(block.shape) == (H, H):
            _PLAN.register(block.detach(), prec, transposed)      # (plan=False: a per-step temporary — its address changes every step)
    if transposed:
        block = block.detach().t().contiguous()
    assert block.shape == (H, H) and block.stride(1) == 1 and block.dtype == torch.float32
    img = torch.empty(H * H, dtype=torch.float32, device=block.device)
    if use_cache:
        _IMG_CACHE[key] = img
    if prec == 2:                                              # 32 KiB bf16 image in the first half of the buffer
        hip.check(hip.lib().namp_pack_image_bf16(block.data_ptr(), block.stride(0), 0, img.data_ptr(), hip.current_stream()),
                  "pack_image_bf16")
    elif prec == 1:
        hip.check(hip.lib().namp_pack_image_x3(block.data_ptr(), block.stride(0), 0, img.data_ptr(), hip.current_stream()),
                  "pack_image_x3")
    else:
        hip.check(hip.lib().namp_pack_image(block.data_ptr(), block.stride(0), 0, H, H, img.data_ptr(), hip.current_stream()),
                  "pack_image")
    return img


def _image_f32(block):
    img = torch.empty(H * H, dtype=torch.float32, device=block.device)
    hip.check(hip.lib().namp_pack_image(block.data_ptr(), block.stride(0), 0, H, H, img.data_ptr(), hip.current_stream()),
              "pack_image")
    return img


def _image_t(block, x3=None, step=None):
    return _image(block.detach(), x3, transposed=True, step=step)


def _reduce(*segs):
    """Sums over partials in ONE HIP launch (namp_reduce_sum): segs = (src, A, Mb, sa, sn, n) with element strides into the contiguous fp32
    tensor `src`  ->  list of [A, Mb] tensors, out[a, b] = sum_{i < n} src[a * sa + i * sn + b].  Up to 16 segments per launch."""
    outs = [torch.empty(A, Mb, device=src.device) for src, A, Mb, sa, sn, n in segs]
    for q0 in range(0, len(segs), 16):
        grp = segs[q0:q0 + 16]
        arr = (hip.NampReduce * len(grp))(*[hip.NampReduce(src.data_ptr(), outs[q0 + i].data_ptr(), A, Mb, sa, sn, n, 0)
                                            for i, (src, A, Mb, sa, sn, n) in enumerate(grp)])
        hip.check(hip.lib().namp_reduce_sum(arr, len(grp), hip.current_stream()), "reduce_sum")
    return outs


def _seg0(t):
    """segment: sum of the contiguous tensor t over its leading dimension -> [1, t[0].numel()]"""
    assert t.is_contiguous() and t.dtype == torch.float32
    M = t[0].numel()
    return (t, 1, M, 0, M, t.shape[0])


def _seg1(t):
    """segment: sum of the contiguous tensor t [A, n, ...] over dimension 1 -> [A, prod(...)]"""
    assert t.is_contiguous() and t.dtype == torch.float32
    M = t[0, 0].numel()
    return (t, t.shape[0], M, t.shape[1] * M, M, t.shape[1])


def _wgrad(G, A, gelu_A, want_bias, x3=None):
    """sum over rows of G^T act(A) (and of G): [128,128] (, [128])."""
    L = hip.lib()
    rows = G.shape[0]
    n = L.namp_train_wgrad_chunks(rows)
    dW = torch.empty(n, H, H, device=G.device)
    db = torch.empty(n, H, device=G.device) if want_bias else None
    hip.check(L.namp_train_wgrad(G.data_ptr(), A.data_ptr(), int(gelu_A), (0 if gelu_A else int(X3 if x3 is None else x3)), rows, dW.data_ptr(), hip.ptr(db),
                                 hip.current_stream()), "train_wgrad")
    if want_bias:
        dW_, db_ = _reduce(_seg0(dW), _seg0(db))
        return dW_.view(H, H), db_.view(H)
    return _reduce(_seg0(dW))[0].view(H, H), None


def _wgrad_many(pairs, x3=None, more=()):
    """Several row contractions over the SAME rows in one go: pairs = [(G, A, want_bias), ...] -> [(dW, db or None), ...].
    The per-chunk partials of all of them land in one buffer and are reduced by ONE sum (a stage's three weight gradients
    otherwise cost six small reductions).  more: an iterable of further lists of the same pairs over OTHER rows (a batch walked in slices,
    the row tensors of one slice alive at a time — consumed lazily, between the launches): they add to the first list's partials (fp32
    rows, precision 1 / 2)."""
    L = hip.lib()
    rows = pairs[0][0].shape[0]
    n = L.namp_train_wgrad_chunks(rows)
    k = len(pairs)
    if more:                                   # slices: all launches on the same chunk count, k * n workgroups = one round of the chip's 512
        n = max(1, min(n, 512 // min(k, 8)))
    dev = pairs[0][0].device
    prec = int(X3 if x3 is None else x3)
    tmp = torch.empty(k, n, H, H, device=dev)
    tmpb = torch.empty(k, n, H, device=dev)
    multi = prec in (1, 2) and all(G.dtype == torch.float32 and A.dtype == torch.float32 for G, A, _ in pairs)
    assert multi or not more, "slices are added up by namp_train_wgrad_multi only"
    if multi:
        # fp32 row tensors, split-bf16 / bf16 products: up to 8 contractions per launch
        arr = lambda ptrs: (C.c_void_p * len(ptrs))(*ptrs)
        for s, sl in enumerate(itertools.chain((pairs,), more)):     # lazily: a generator may launch the producer of slice s when asked for it
            r = sl[0][0].shape[0]
            assert len(sl) == k
            for q0 in range(0, k, 8):
                grp = sl[q0:q0 + 8]
                hip.check(L.namp_train_wgrad_multi(arr([G.data_ptr() for G, _, _ in grp]), arr([A.data_ptr() for _, A, _ in grp]), len(grp),
                                                   prec | (64 if s else 0), r, n, arr([tmp[q0 + i].data_ptr() for i in range(len(grp))]),
                                                   arr([(tmpb[q0 + i].data_ptr() if grp[i][2] else None) for i in range(len(grp))]),
                                                   hip.current_stream()), "train_wgrad_multi")
    for q, (G, A, wb) in enumerate(() if multi else pairs):
        assert G.shape[0] == rows
        code = prec
        if G.dtype == torch.bfloat16:                       # the mixed-precision backward's bf16 row tensors
            code |= 16 | (32 if A.dtype == torch.bfloat16 else 0)
        elif A.dtype == torch.bfloat16:
            A = A.float()
        hip.check(L.namp_train_wgrad(G.data_ptr(), A.data_ptr(), 0, code, rows, tmp[q].data_ptr(), tmpb[q].data_ptr() if wb else None,
                                     hip.current_stream()), "train_wgrad")
    if any(wb for _, _, wb in pairs):
        dW, db = _reduce(_seg1(tmp.view(k, n, H * H)), _seg1(tmpb))
    else:
        dW, db = _reduce(_seg1(tmp.view(k, n, H * H)))[0], None
    dW = dW.view(k, H, H)
    return [(dW[q], (db[q] if wb else None)) for q, (_, _, wb) in enumerate(pairs)]


class ReverseAdjacency:
    """Edges grouped by the table row they gather (global row b*N + E_idx[b,i,k]) — the transpose of the neighbour
    gather, built once per step (namp_train_reverse_adjacency: a counting sort on the device) and shared by all per-edge stages'
    backward passes."""

    def __init__(self, E_idx32):
        B, N, K = E_idx32.shape
        dev = E_idx32.device
        E_idx32 = E_idx32.contiguous()
        self.G, self.K, self.N = B * N, K, N
        self.edges = torch.empty(B * N * K, dtype=torch.int32, device=dev)
        self.offsets = torch.empty(B * N + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(2 * B * N + B * N * K, dtype=torch.int32, device=dev)
        hip.check(hip.lib().namp_train_reverse_adjacency(E_idx32.data_ptr(), self.offsets.data_ptr(), self.edges.data_ptr(), ws.data_ptr(),
                                                         B, N, K, hip.current_stream()), "train_reverse_adjacency")
        self._E_idx32 = E_idx32
        self._sel = {}

    def decoder_sel(self, rank32):
        """uint8 per edge: 1 where the gathered neighbour comes EARLIER in the decoding order (its row was read from the first table).  The same for
        all decoder layers of a step: computed once per rank tensor."""
        key = (rank32.data_ptr(), rank32._version)
        hit = self._sel.get(key)
        if hit is None:
            B = self.G // self.N
            r = rank32.view(B, self.N)
            rj = torch.gather(r, 1, self._E_idx32.view(B, -1).long()).view(B, self.N, self.K)
            hit = (rj < r[:, :, None]).to(torch.uint8).contiguous().view(-1)
            self._sel = {key: hit}
        return hit

    def scatter(self, G1, sel=None):
        """-> sum of G1 rows per gathered table row [G,128] (two outputs when sel, uint8 per edge, is given)."""
        out0 = torch.empty(self.G, H, device=G1.device)
        out1 = torch.empty(self.G, H, device=G1.device) if sel is not None else None
        fn = hip.lib().namp_train_scatter_rows_bf16 if G1.dtype == torch.bfloat16 else hip.lib().namp_train_scatter_rows
        hip.check(fn(G1.data_ptr(), self.edges.data_ptr(), self.offsets.data_ptr(), hip.ptr(sel),
                     out0.data_ptr(), hip.ptr(out1), self.G, hip.current_stream()), "scatter_rows")
        return out0, out1


class _EdgeMLP(torch.autograd.Function):
    """One per-edge 3-layer MLP of EncLayer / DecLayer with the hoisted first layer
    ``z1 = W1b.h_E[i,k] + Pa[i] + Pj[j]``.  mode 0/1 -> sum_k w_ik * MLP / 30 per residue; mode 2 -> the message per edge."""

    @staticmethod
    def forward(ctx, mode, h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, b3, E_idx32, mask32, mask_attend32, rank32, rev=None):
        B, N, K = E_idx32.shape
        L = hip.lib()
        h_E, Pa, Pj0 = h_E.contiguous(), Pa.contiguous(), Pj0.contiguous()
        Pj1 = Pj1.contiguous() if Pj1 is not None else None
        imgs = [_image(W1b.detach()), _image(W2.detach()), _image(W3.detach())]
        b2c, b3c = b2.detach().contiguous(), b3.detach().contiguous()
        G, tpn = B * N, (K + 15) // 16
        if mode == ENC_EDGE:
            out = torch.empty(B, N, K, H, device=h_E.device)
        else:
            # message modes: K-sums of the layer-2 activations per 16-neighbour tile + the tiles' weight sums (include/namp.h);
            # layer 3 is linear and is applied to the summed rows below — once per residue instead of once per edge
            out = torch.empty(G * tpn * (H + 1) + 3, device=h_E.device)
        hip.check(L.namp_train_edge_fwd(mode, h_E.data_ptr(), E_idx32.data_ptr(), hip.ptr(mask32), hip.ptr(mask_attend32),
                                        hip.ptr(rank32), Pa.data_ptr(), Pj0.data_ptr(), hip.ptr(Pj1), imgs[0].data_ptr(),
                                        imgs[1].data_ptr(), imgs[2].data_ptr(), b2c.data_ptr(), b3c.data_ptr(),
                                        None, None, 0.0, 0, out.data_ptr(), int(X3), B, N, K, hip.current_stream()), "train_edge_fwd")
        ctx.mode, ctx.rev, ctx.x3, ctx.step = mode, rev, X3, _STEP     # backward runs at the precision of ITS forward
        ctx.set_materialize_grads(False)
        if mode == ENC_EDGE:
            ctx.save_for_backward(h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, E_idx32, mask32, mask_attend32, rank32)
            return out
        # (the launch's per-tile K-sums and weight sums -> per residue: one reduction launch)
        msum, wsum = _reduce((out, G, H, tpn * H, H, tpn), (out[G * tpn * H:], G, 1, tpn, 1, tpn))
        ctx.save_for_backward(h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, E_idx32, mask32, mask_attend32, rank32, msum, wsum)
        # layer 3 behind the K-sum, per residue: W3 . msum + b3 * wsum — node_linear_kernel (fp32-equivalent split-bf16 products in both the
        # split-bf16 and the mixed-precision mode: this is residue-level math), not the stock GEMM
        dh = torch.addcmul(_node_linear_call(msum, [W3.detach()], [None], x3=min(int(X3), 1))[0], wsum, b3.detach())
        return dh.view(B, N, H), h_E.view_as(h_E)
        # (second output: h_E itself, for the NEXT consumer of the same edge rows (EncLayer's edge update, the next DecLayer).
        # Its gradient then arrives HERE, and the backward launch writes (that gradient + its own dL/dh_E) in one pass — autograd
        # would otherwise sum the consumers' [E,128] gradients with a separate 1.8 GB pass each.)

    @staticmethod
    def backward(ctx, g, g_pass=None):
        mode = ctx.mode
        if mode == ENC_EDGE:
            h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, E_idx32, mask32, mask_attend32, rank32 = ctx.saved_tensors
        else:
            h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, E_idx32, mask32, mask_attend32, rank32, msum, wsum = ctx.saved_tensors
        B, N, K = E_idx32.shape
        E = B * N * K
        dev = h_E.device
        L = hip.lib()
        if g is None:                                        # only the pass-through output was used downstream
            g = torch.zeros(B, N, H, device=dev) if mode != ENC_EDGE else torch.zeros(B, N, K, H, device=dev)
        g = g.contiguous()
        img1, img2 = _image(W1b.detach(), ctx.x3, step=ctx.step), _image(W2.detach(), ctx.x3, step=ctx.step)
        img2t, img1t = _image_t(W2, ctx.x3, ctx.step), _image_t(W1b, ctx.x3, ctx.step)
        img3t = _image_t(W3, ctx.x3, ctx.step) if mode == ENC_EDGE else None
        if mode != ENC_EDGE:
            # layer 3 behind the K-sum: its gradients are residue-level products, and the edge kernel gets dL/d(K-sum)
            g2d = g.view(B * N, H)
            # dW3 = g^T . (K-sums): a 24,000-row contraction — the row-contraction kernel (the library GEMM picks a 32 x 32 x 64 tiling
            # for this [128 x 24,000] x [24,000 x 128] shape: 85 us against ~30)
            dW3 = _wgrad_many([(g2d.contiguous(), msum.contiguous(), False)], x3=ctx.x3)[0][0]
            g2c = g2d.contiguous()
            n3 = L.namp_train_rows_groups(g2c.shape[0])
            p3 = torch.empty(n3, H, device=dev)
            hip.check(L.namp_train_wcolsum(g2c.data_ptr(), wsum.contiguous().data_ptr(), g2c.shape[0], p3.data_ptr(), hip.current_stream()),
                      "train_wcolsum")
            db3 = _reduce(_seg0(p3))[0].view(H)
            g = _node_linear_call(g2d.contiguous(), [W3.detach()], [None], x3=min(int(ctx.x3), 1), transposed=True, step=ctx.step)[0]
        rdt = torch.bfloat16 if int(ctx.x3) == 2 else torch.float32          # mixed precision: bf16 row tensors
        if mode != ENC_EDGE and int(ctx.x3) in (1, 2) and DW_ONCHIP:
            return _EdgeMLP._backward_dw(ctx, g, g_pass, dW3, db3)
        A1, G1, G2 = (torch.empty(E, H, device=dev, dtype=rdt) for _ in range(3))
        # the later consumer's dL/dh_E arrives as g_pass: the launch reads it and writes the SUM to a fresh buffer (same HBM
        # traffic as adding in place — one read and one write of [E,128] — without mutating a gradient autograd handed in,
        # which a hook / retain_grad / a second consumer of the pass-through output could still be looking at)
        acc = g_pass is not None
        if acc and not (g_pass.is_contiguous() and g_pass.dtype == torch.float32):
            g_pass = g_pass.contiguous().float()
        g_hE = torch.empty(E, H, device=dev)
        A2 = torch.empty(E, H, device=dev, dtype=rdt) if mode == ENC_EDGE else None
        G3 = S3 = w3 = None
        b2c = b2.detach().contiguous()
        gpa_tiles = K % 16 == 0                                # deterministic per-tile sums instead of fp32 atomics
        g_Pa = torch.empty(E // 16, H, device=dev) if gpa_tiles else torch.zeros(B * N, H, device=dev)
        hip.check(L.namp_train_edge_bwd(mode, h_E.data_ptr(), E_idx32.data_ptr(), hip.ptr(mask32), hip.ptr(mask_attend32),
                                        hip.ptr(rank32), Pa.data_ptr(), Pj0.data_ptr(), hip.ptr(Pj1), img1.data_ptr(),
                                        img2.data_ptr(), hip.ptr(img3t), img2t.data_ptr(), img1t.data_ptr(), b2c.data_ptr(),
                                        g.data_ptr(), A1.data_ptr(), hip.ptr(A2), G1.data_ptr(), G2.data_ptr(), hip.ptr(G3),
                                        g_hE.data_ptr(), (g_pass.data_ptr() if acc else None), g_Pa.data_ptr(), None, None, hip.ptr(S3), hip.ptr(w3),
                                        int(ctx.x3) | (4 if acc else 0) | (8 if gpa_tiles else 0),
                                        B, N, K, hip.current_stream()), "train_edge_bwd")
        rev = ctx.rev if ctx.rev is not None else ReverseAdjacency(E_idx32)
        if mode == DEC_MSG:
            g_Pj0, g_Pj1 = rev.scatter(G1, rev.decoder_sel(rank32))
        else:
            g_Pj0, g_Pj1 = rev.scatter(G1)
        if mode == ENC_EDGE:
            G3 = g.view(E, H).to(rdt)
            (dW3, db3), (dW2, db2), (dW1b, _) = _wgrad_many([(G3, A2, True), (G2, A1, True), (G1, h_E.view(E, H), False)], x3=ctx.x3)
        else:
            (dW2, db2), (dW1b, _) = _wgrad_many([(G2, A1, True), (G1, h_E.view(E, H), False)], x3=ctx.x3)
        if gpa_tiles:
            g_Pa = _reduce((g_Pa, B * N, H, (K // 16) * H, H, K // 16))[0]
        g_Pa, g_Pj0 = g_Pa.view_as(Pa), g_Pj0.view_as(Pj0)
        g_Pj1 = g_Pj1.view_as(Pj1) if g_Pj1 is not None else None
        return (None, g_hE.view_as(h_E), g_Pa, g_Pj0, g_Pj1, dW1b, dW2, db2, dW3, db3, None, None, None, None, None)


    @staticmethod
    def _backward_dw(ctx, g, g_pass, dW3, db3):
        """Message stages, split-bf16 / bf16 products: ONE persistent launch that also contracts (G2, A1) and (G1, h_E) over the edges
        on chip (csrc/namp_train_dw.h) — no A1 / G2 row tensors, no row-contraction launches.  g = dL/d(K-sum) per residue."""
        h_E, Pa, Pj0, Pj1, W1b, W2, b2, W3, E_idx32, mask32, mask_attend32, rank32, msum, wsum = ctx.saved_tensors
        mode = ctx.mode
        B, N, K = E_idx32.shape
        E = B * N * K
        dev = h_E.device
        L = hip.lib()
        img1, img2 = _image(W1b.detach(), ctx.x3, step=ctx.step), _image(W2.detach(), ctx.x3, step=ctx.step)
        img2t, img1t = _image_t(W2, ctx.x3, ctx.step), _image_t(W1b, ctx.x3, ctx.step)
        rdt = torch.bfloat16 if int(ctx.x3) == 2 else torch.float32
        Ep = L.namp_train_edge_bwd_dw_rows(B, N, K)            # whole 64-row rounds: the launch stores rows past E into this padding
        G1 = torch.empty(Ep, H, device=dev, dtype=rdt)[:E]
        acc = g_pass is not None
        if acc and not (g_pass.is_contiguous() and g_pass.dtype == torch.float32):
            g_pass = g_pass.contiguous().float()
        g_hE = torch.empty(Ep, H, device=dev)[:E]
        b2c = b2.detach().contiguous()
        gpa_tiles = K % 16 == 0
        g_Pa = torch.empty(Ep // 16, H, device=dev)[:E // 16] if gpa_tiles else torch.zeros(B * N, H, device=dev)
        n = L.namp_train_edge_bwd_dw_groups(B, N, K)
        dWp = torch.empty(n, 2, H, H, device=dev)
        dbp = torch.empty(n, H, device=dev)
        if mode == ENC_MSG and mask32 is None and mask_attend32 is None:
            mask32 = torch.ones(B, N, dtype=torch.int32, device=dev)       # "no mask" = all ones (the persistent launch reads one unconditionally)
        hip.check(L.namp_train_edge_bwd_dw(mode, h_E.data_ptr(), E_idx32.data_ptr(), hip.ptr(mask32), hip.ptr(mask_attend32),
                                           hip.ptr(rank32), Pa.data_ptr(), Pj0.data_ptr(), hip.ptr(Pj1), img1.data_ptr(),
                                           img2.data_ptr(), img2t.data_ptr(), img1t.data_ptr(), b2c.data_ptr(), g.data_ptr(),
                                           G1.data_ptr(), g_hE.data_ptr(), (g_pass.data_ptr() if acc else None), g_Pa.data_ptr(),
                                           dWp.data_ptr(), dbp.data_ptr(), int(ctx.x3) | (4 if acc else 0) | (8 if gpa_tiles else 0),
                                           B, N, K, hip.current_stream()), "train_edge_bwd_dw")
        rev = ctx.rev if ctx.rev is not None else ReverseAdjacency(E_idx32)
        if mode == DEC_MSG:
            g_Pj0, g_Pj1 = rev.scatter(G1, rev.decoder_sel(rank32))
        else:
            g_Pj0, g_Pj1 = rev.scatter(G1)
        # the workgroups' partials and the per-tile dL/dPa rows: one reduction launch
        segs = [_seg0(dWp), _seg0(dbp)] + ([(g_Pa, B * N, H, (K // 16) * H, H, K // 16)] if gpa_tiles else [])
        red = _reduce(*segs)
        dW2, dW1b = red[0].view(2, H, H)
        db2 = red[1].view(H)
        if gpa_tiles:
            g_Pa = red[2]
        g_Pa, g_Pj0 = g_Pa.view_as(Pa), g_Pj0.view_as(Pj0)
        g_Pj1 = g_Pj1.view_as(Pj1) if g_Pj1 is not None else None
        return (None, g_hE.view_as(h_E), g_Pa, g_Pj0, g_Pj1, dW1b, dW2, db2, dW3, db3, None, None, None, None, None)


class _EdgeUpdate(torch.autograd.Function):
    """The whole EncLayer edge update h_E' = LayerNorm3(h_E + dropout3(MLP'([h_V_i | h_E | h_V_j])))
    (na_model_utils.py:236-240) as one forward and one backward launch: the message never goes to memory, the dropout mask
    is a counter-based hash regenerated in backward, LayerNorm3 is differentiated in registers."""

    @staticmethod
    def forward(ctx, h_E, Pa, Pc, W1b, W2, b2, W3, b3, ln_w, ln_b, E_idx32, p, seed, rev=None):
        B, N, K = E_idx32.shape
        h_E, Pa, Pc = h_E.contiguous(), Pa.contiguous(), Pc.contiguous()
        imgs = [_image(W1b.detach()), _image(W2.detach()), _image(W3.detach())]
        b2c, b3c = b2.detach().contiguous(), b3.detach().contiguous()
        g_, b_ = ln_w.detach().contiguous(), ln_b.detach().contiguous()
        out = torch.empty_like(h_E)
        hip.check(hip.lib().namp_train_edge_fwd(ENC_EDGE, h_E.data_ptr(), E_idx32.data_ptr(), None, None, None, Pa.data_ptr(),
                                                Pc.data_ptr(), None, imgs[0].data_ptr(), imgs[1].data_ptr(), imgs[2].data_ptr(),
                                                b2c.data_ptr(), b3c.data_ptr(), g_.data_ptr(), b_.data_ptr(), float(p), int(seed),
                                                out.data_ptr(), int(X3), B, N, K, hip.current_stream()), "train_edge_fwd")
        ctx.p, ctx.seed, ctx.rev, ctx.x3, ctx.step = float(p), int(seed), rev, X3, _STEP
        ctx.save_for_backward(h_E, Pa, Pc, W1b, W2, b2, W3, b3, ln_w, E_idx32)
        return out

    @staticmethod
    def backward(ctx, g):
        h_E, Pa, Pc, W1b, W2, b2, W3, b3, ln_w, E_idx32 = ctx.saved_tensors
        B, N, K = E_idx32.shape
        E = B * N * K
        dev = h_E.device
        L = hip.lib()
        g = g.contiguous()
        img1, img2, img3 = (_image(W1b.detach(), ctx.x3, step=ctx.step), _image(W2.detach(), ctx.x3, step=ctx.step),
                            _image(W3.detach(), ctx.x3, step=ctx.step))
        img3t, img2t, img1t = _image_t(W3, ctx.x3, ctx.step), _image_t(W2, ctx.x3, ctx.step), _image_t(W1b, ctx.x3, ctx.step)
        rdt = torch.bfloat16 if int(ctx.x3) == 2 else torch.float32          # mixed precision: bf16 row tensors
        if int(ctx.x3) == 2 and DW_ONCHIP and DW_ONCHIP_EDGE:
            return _EdgeUpdate._backward_dw(ctx, g, (img1, img2, img3, img3t, img2t, img1t))
        # Row tensors A1, A2, G2, G3 exist for the contractions only.  Split-bf16 (fp32 rows, 590 MB each at cfg5): the batch is walked in
        # two slices of complexes — launch, contract (adding to the first slice's partials), next slice into the same buffers — so that one
        # slice's rows are alive at a time: 6.7 -> 5.6 GiB peak at cfg5 (+0.2 ms).  G1 (the table-gradient gather reads it through the reverse adjacency
        # of the whole batch), dL/dh_E and the per-tile dL/dPa rows are whole-batch tensors the slices fill.
        nsl = min(B, EDGE_UPDATE_SLICES) if int(ctx.x3) == 1 else 1
        bounds = [i * (B // nsl) + min(i, B % nsl) for i in range(nsl + 1)]        # the larger slices first (the first one sizes the partials)
        Es = max(b1 - b0 for b0, b1 in zip(bounds, bounds[1:])) * N * K
        A1, A2, G2, G3 = (torch.empty(Es, H, device=dev, dtype=rdt) for _ in range(4))
        G1 = torch.empty(E, H, device=dev, dtype=rdt)
        g_hE = torch.empty(E, H, device=dev)
        gpa_tiles = K % 16 == 0
        g_Pa = torch.empty(E // 16, H, device=dev) if gpa_tiles else torch.zeros(B * N, H, device=dev)
        groups = [L.namp_train_edge_update_bwd_groups(b1 - b0, N, K) for b0, b1 in zip(bounds, bounds[1:])]
        part = torch.empty(sum(groups), 2, H, device=dev)
        b2c, b3c, lw = b2.detach().contiguous(), b3.detach().contiguous(), ln_w.detach().contiguous()
        hE2, g2, Pa2, Pc2 = h_E.view(E, H), g.view(E, H), Pa.view(B * N, H), Pc.view(B * N, H)

        def launch(i):
            b0, b1 = bounds[i], bounds[i + 1]
            e0, e1, n0 = b0 * N * K, b1 * N * K, b0 * N
            hip.check(L.namp_train_edge_update_bwd(hE2[e0:].data_ptr(), E_idx32[b0:].data_ptr(), Pa2[n0:].data_ptr(), Pc2[n0:].data_ptr(),
                                                   img1.data_ptr(), img2.data_ptr(), img3.data_ptr(), img3t.data_ptr(), img2t.data_ptr(),
                                                   img1t.data_ptr(), b2c.data_ptr(), b3c.data_ptr(), lw.data_ptr(), ctx.p, ctx.seed, e0,
                                                   g2[e0:].data_ptr(), A1.data_ptr(), A2.data_ptr(), G1[e0:].data_ptr(), G2.data_ptr(),
                                                   G3.data_ptr(), g_hE[e0:].data_ptr(), (g_Pa[e0 // 16:] if gpa_tiles else g_Pa[n0:]).data_ptr(),
                                                   None, part[sum(groups[:i]):].data_ptr(), int(ctx.x3) | (8 if gpa_tiles else 0), b1 - b0, N, K,
                                                   hip.current_stream()), "train_edge_update_bwd")
            r = e1 - e0
            return [(G3[:r], A2[:r], True), (G2[:r], A1[:r], True), (G1[e0:e1], hE2[e0:e1], False)]

        (dW3, db3), (dW2, db2), (dW1b, _) = _wgrad_many(launch(0), x3=ctx.x3, more=(launch(i) for i in range(1, nsl)) if nsl > 1 else ())
        red = _reduce(*([_seg0(part)] + ([(g_Pa, B * N, H, (K // 16) * H, H, K // 16)] if gpa_tiles else [])))
        dgb = red[0].view(2, H)
        if gpa_tiles:
            g_Pa = red[1]
        rev = ctx.rev if ctx.rev is not None else ReverseAdjacency(E_idx32)
        g_Pc, _ = rev.scatter(G1)
        return (g_hE.view_as(h_E), g_Pa.view_as(Pa), g_Pc.view_as(Pc), dW1b, dW2, db2, dW3, db3, dgb[0], dgb[1],
                None, None, None, None)


def _edge_update_backward_dw(ctx, g, imgs):
    """Mixed precision: two persistent launches (csrc/namp_train_eu.h) that contract dW3 / db3 (launch A, with the LayerNorm and dropout backward) and
    dW2 / db2, dW1b (launch B) over the edges on chip; between them only the bf16 rows G2 = dL/dz2 and the fp32 rows dL/dx (parked in g_hE)."""
    h_E, Pa, Pc, W1b, W2, b2, W3, b3, ln_w, E_idx32 = ctx.saved_tensors
    img1, img2, img3, img3t, img2t, img1t = imgs
    B, N, K = E_idx32.shape
    E = B * N * K
    dev = h_E.device
    L = hip.lib()
    Ep = L.namp_train_edge_bwd_dw_rows(B, N, K)
    G2, G1 = (torch.empty(Ep, H, device=dev, dtype=torch.bfloat16) for _ in range(2))
    g_hE = torch.empty(Ep, H, device=dev)[:E]
    gpa_tiles = K % 16 == 0
    g_Pa = torch.empty(Ep // 16, H, device=dev)[:E // 16] if gpa_tiles else torch.zeros(B * N, H, device=dev)
    n = L.namp_train_edge_bwd_dw_groups(B, N, K)
    dWp = torch.empty(3 * n, H, H, device=dev)               # [n] dW3 partials, then [n][2] (dW2, dW1b)
    dbp = torch.empty(2 * n, H, device=dev)                  # [n] db3, then [n] db2
    part = torch.empty(n, 2, H, device=dev)
    b2c, b3c, lw = b2.detach().contiguous(), b3.detach().contiguous(), ln_w.detach().contiguous()
    hip.check(L.namp_train_edge_update_bwd_dw(h_E.data_ptr(), E_idx32.data_ptr(), Pa.data_ptr(), Pc.data_ptr(), img1.data_ptr(),
                                              img2.data_ptr(), img3.data_ptr(), img3t.data_ptr(), img2t.data_ptr(), img1t.data_ptr(),
                                              b2c.data_ptr(), b3c.data_ptr(), lw.data_ptr(), ctx.p, ctx.seed, g.data_ptr(),
                                              G2.data_ptr(), G1.data_ptr(), g_hE.data_ptr(), g_Pa.data_ptr(),
                                              dWp.data_ptr(), dbp.data_ptr(), part.data_ptr(), 2 | (8 if gpa_tiles else 0),
                                              B, N, K, hip.current_stream()), "train_edge_update_bwd_dw")
    rev = ctx.rev if ctx.rev is not None else ReverseAdjacency(E_idx32)
    g_Pc, _ = rev.scatter(G1[:E])
    # the two launches' partials and the per-tile dL/dPa rows: one reduction launch
    segs = [_seg0(dWp[:n]), _seg0(dWp[n:].view(n, 2 * H * H)), _seg1(dbp.view(2, n, H)), _seg0(part)] + \
        ([(g_Pa, B * N, H, (K // 16) * H, H, K // 16)] if gpa_tiles else [])
    red = _reduce(*segs)
    dW3, dW21, db, dgb = red[0].view(H, H), red[1].view(2, H, H), red[2], red[3].view(2, H)
    if gpa_tiles:
        g_Pa = red[4]
    return (g_hE.view_as(h_E), g_Pa.view_as(Pa), g_Pc.view_as(Pc), dW21[1], dW21[0], db[1], dW3, db[0], dgb[0], dgb[1],
            None, None, None, None)


_EdgeUpdate._backward_dw = staticmethod(_edge_update_backward_dw)


class _TableRows(torch.autograd.Function):
    """table[idx] for a table with FEW rows and 128 columns (the 6 polymer types of node_embedding, the 33 tokens of W_s).  The stock
    index / embedding backward scatters the 24,000 rows into those few with atomics; here the gradient is the per-class sum on
    namp_train_class_sums (per-wave private tables in LDS: deterministic) + one namp_reduce_sum."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.nrows = table.shape[0]
        return table[idx]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        C_ = g.shape[-1]
        g2 = g.reshape(-1, C_).contiguous().float()
        i32 = idx.reshape(-1).to(torch.int32).contiguous()
        rows = g2.shape[0]
        L = hip.lib()
        if not g2.is_cuda:
            raise RuntimeError("na_mpnn_amd.train: tensors must be on a HIP device (no CPU fallback)")
        if C_ != H or ctx.nrows > 64:                         # not a table of this model: the stock scatter
            return torch.zeros(ctx.nrows, C_, device=g.device, dtype=g2.dtype).index_add_(0, i32.long(), g2), None
        n = L.namp_train_rows_groups(rows)
        part = torch.empty(n, ctx.nrows, H, device=g.device)
        hip.check(L.namp_train_class_sums(g2.data_ptr(), i32.data_ptr(), ctx.nrows, rows, part.data_ptr(), hip.current_stream()), "train_class_sums")
        return _reduce(_seg0(part))[0].view(ctx.nrows, H), None


class _EdgeLinear(torch.autograd.Function):
    """y = x W^T + b over [B,N,K,128] edge rows (W_e, na_model_utils.py:598) on the single-GEMM mode of the edge kernel;
    backward: dL/dx with the image of W^T, dW / db with the row-contraction kernel (the library GEMM spends 2 ms on this
    128 x 128 x 10^6 shape)."""

    @staticmethod
    def forward(ctx, x, W, b):
        B, N, K, _ = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        hip.check(hip.lib().namp_edge_embed_prec(_image(W.detach()).data_ptr(), b.detach().contiguous().data_ptr(), x.data_ptr(),
                                                 y.data_ptr(), int(X3), B, N, K, hip.current_stream()), "edge_embed")
        ctx.x3, ctx.step = X3, _STEP
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        B, N, K, _ = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x)
        zero = torch.zeros(H, device=x.device)
        hip.check(hip.lib().namp_edge_embed_prec(_image_t(W, ctx.x3, ctx.step).data_ptr(), zero.data_ptr(), g.data_ptr(), gx.data_ptr(), int(ctx.x3),
                                                 B, N, K, hip.current_stream()), "edge_embed (dgrad)")
        dW, db = _wgrad(g.view(-1, H), x.view(-1, H), False, True, x3=ctx.x3)
        return gx, dW, db


def _atom_frames(model, X, fd):
    """X18 [B,L,18,3] and M18 [B,L,18] (na_model_utils.py:472-497): 16 atoms + virtual Cb + virtual N_na."""
    ad = model.atom_dict
    Ca = X[:, :, ad["CA"]]
    Cb = model._virtual(X[:, :, ad["N"]], Ca, X[:, :, ad["C"]], -0.58273431, 0.56802827, -0.54067466)
    C1p = X[:, :, ad["C1'"]]
    Nna = model._virtual(X[:, :, ad["O4'"]], C1p, X[:, :, ad["C2'"]], -0.56967352, 0.51055973, -0.53122153)
    X18 = torch.cat((X, Cb[:, :, None], Nna[:, :, None]), -2).contiguous()
    dna_m, rna_m = model._na_masks(fd)                         # all-zero when the model has no virtual N_na atom
    M18 = torch.cat((fd["X_m"], fd["protein_mask"][:, :, None], (rna_m + dna_m)[:, :, None]), -1)
    return X18, M18.float().contiguous()


def _edge_embedding_fwd(fp, W18, X, ints, top_k, ref_atom):
    """Fused HIP featuriser without its LayerNorm: y = edge_embedding([E_pos | RBF]) [B,L,K,128] and E_idx int32
    (na_model_utils.py:489-507); no autograd here, see _EdgeEmbeddingGrad."""
    L = hip.lib()
    B, Lr = X.shape[:2]
    K = int(min(top_k, Lr))
    dev = X.device
    Wd = W18.detach().contiguous()                            # 18-atom column layout [128 x 5200] (model.edge_weight18)
    img = torch.empty(Wd.numel(), device=dev)
    m = hip.NampModelW()
    if X3:                                                    # the feature GEMM: split products in both reduced-precision modes
        hip.check(L.namp_pack_feat_x3(Wd.data_ptr(), Wd.shape[1], img.data_ptr(), hip.current_stream()), "pack_feat_x3(edge_embedding)")
        m.feat.Wedge_ximg = img.data_ptr()
        m.reserved = 2 if X3 == 2 else 0                       # mixed precision: plain bf16 products on the hi half
    else:
        hip.check(L.namp_pack_image(Wd.data_ptr(), Wd.shape[1], 0, H, Wd.shape[1], img.data_ptr(), hip.current_stream()),
                  "pack_image(edge_embedding)")
        m.feat.Wedge_img = img.data_ptr()
    pw, pb = fp.embeddings.linear.weight.detach().contiguous(), fp.embeddings.linear.bias.detach().contiguous()
    m.feat.pos_w, m.feat.pos_b = pw.data_ptr(), pb.data_ptr()
    m.feat.ln_g = m.feat.ln_b = None                          # pre-LayerNorm rows
    E_idx = torch.empty(B, Lr, K, dtype=torch.int32, device=dev)
    y = torch.empty(B, Lr, K, H, device=dev)
    ws = torch.empty(L.namp_featurize_workspace_bytes(B, Lr), dtype=torch.uint8, device=dev)
    hip.check(L.namp_featurize(C.byref(m), X.data_ptr(), *[t.data_ptr() for t in ints], int(top_k), int(ref_atom),
                               E_idx.data_ptr(), y.data_ptr(), None, ws.data_ptr(), ws.numel(), B, Lr,
                               hip.current_stream()), "featurize")
    return y, E_idx


class _EdgeEmbeddingGrad(torch.autograd.Function):
    """Identity on y that routes dL/dy into edge_embedding.weight and the positional embedding (embeddings.linear weight / bias): the weight
    gradient of the 5200 -> 128 embedding (namp_train_feat_wgrad) and the positional table gradient (namp_train_pos_grad: the data gradient
    g . Wedge[:, :16] contracted per relative-position class on chip — no [E,16] tensors, no library GEMM, no one-hot products)."""

    @staticmethod
    def forward(ctx, y, Wedge, pos_w, pos_b, E_pos, d32, X18, M18, E_idx):
        ctx.x3 = X3
        ctx.save_for_backward(Wedge, E_pos, d32, X18, M18, E_idx)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g_y):
        Wedge, E_pos, d32, X18, M18, E_idx = ctx.saved_tensors
        L = hip.lib()
        B, Lr, K = E_idx.shape
        E = B * Lr * K
        g = g_y.contiguous()
        n = L.namp_train_feat_wgrad_chunks(E)
        part = torch.empty(n, H, Wedge.shape[1], device=g.device)
        tws = torch.empty(L.namp_train_feat_wgrad_ws_ints(E), dtype=torch.int32, device=g.device)
        if int(ctx.x3):                                   # packed atoms (x, y, z, mask): one 16-byte request per gathered atom
            XM = torch.cat((X18, M18.unsqueeze(-1)), -1).contiguous()
            xp, mp = XM.data_ptr(), None
        else:
            xp, mp = X18.data_ptr(), M18.data_ptr()
        g16 = _G16.pop(g.data_ptr(), None)                     # bf16 operand tiles of this very tensor, left by _EdgeEmbedTail.backward
        # (same tensor, UNEDITED: a hook that scales or clips dL/dy in place between the two nodes bumps its version counter, and the
        # tiles would then describe the old values while pos_grad below reads the new ones — fall back to the fp32 rows)
        if g16 is not None and (g16[1] != E or g16[2] != int(ctx.x3) or g16[3] != g._version):
            g16 = None
        hip.check(L.namp_train_feat_wgrad(xp, mp, E_idx.data_ptr(), E_pos.data_ptr(), g.data_ptr(), g16[0].data_ptr() if g16 else None,
                                          part.data_ptr(), tws.data_ptr(), int(ctx.x3), B, Lr, K, hip.current_stream()),
                  "train_feat_wgrad")
        _G16.clear()
        Wc = Wedge.detach().contiguous()
        n2 = L.namp_train_pos_grad_groups(E)
        part2 = torch.empty(n2, spec.NUM_POS_CLASSES + 1, spec.NUM_POS, device=g.device)
        hip.check(L.namp_train_pos_grad(g.data_ptr(), Wc.data_ptr(), Wc.stride(0), d32.data_ptr(), part2.data_ptr(), E, hip.current_stream()),
                  "train_pos_grad")
        dW, tab = _reduce(_seg0(part), _seg0(part2))
        tab = tab.view(spec.NUM_POS_CLASSES + 1, spec.NUM_POS)
        return (None, dW.view(part.shape[1:]), tab[:spec.NUM_POS_CLASSES].t().contiguous(), tab[spec.NUM_POS_CLASSES].contiguous(),
                None, None, None, None, None)


def edge_embedding(model, fd):
    """-> y_pre [B,L,K,128] (differentiable w.r.t. features.edge_embedding / features.embeddings), E_idx int32."""
    fp = model.features
    X = model._noised_X(fd).float().contiguous()
    X18, M18 = _atom_frames(model, X, fd)
    dna_m, rna_m = model._na_masks(fd)
    ints = [fd[k].to(torch.int32).contiguous() for k in ("X_m", "mask", "R_idx", "chain_labels", "protein_mask")] + \
        [dna_m.to(torch.int32).contiguous(), rna_m.to(torch.int32).contiguous()]
    W18 = model.edge_weight18()                               # differentiable expansion when include_pred_na_N = 0
    with torch.no_grad():
        y, E_idx = _edge_embedding_fwd(fp, W18, X, ints, model.k_neighbors, model.atom_dict[model.na_ref_atom])
        # positional features (PositionalEncodings, na_model_utils.py:537-541): class index and E_pos rows from one launch; their gradient
        # reaches embeddings.linear through _EdgeEmbeddingGrad
        B, Lr, K = E_idx.shape
        pw, pb = fp.embeddings.linear.weight, fp.embeddings.linear.bias
        d32 = torch.empty(B, Lr, K, dtype=torch.int32, device=X.device)
        E_pos = torch.empty(B, Lr, K, spec.NUM_POS, device=X.device)
        hip.check(hip.lib().namp_train_pos_features(ints[2].data_ptr(), ints[3].data_ptr(), E_idx.data_ptr(), pw.detach().contiguous().data_ptr(),
                                                    pb.detach().contiguous().data_ptr(), d32.data_ptr(), E_pos.data_ptr(), B, Lr, K,
                                                    hip.current_stream()), "train_pos_features")
    y = _EdgeEmbeddingGrad.apply(y, W18, pw, pb, E_pos, d32, X18, M18, E_idx)
    return y, E_idx


def _ln(x, norm):
    return F.layer_norm(x, (H,), norm.weight, norm.bias, 1e-5)


def _node_linear_call(x2, blocks, biases, x3=None, transposed=False, step=None, plan=True):
    """y_q = x2 @ blocks[q]^T (+ biases[q]) — transposed: x2 @ blocks[q] — for up to 8 [128 x 128] blocks in ONE node_linear launch
    -> list of [G,128].  Products at the step's precision: exact fp32 MFMA (code 0), split-bf16 (1), plain bf16 (2: the hi plane of
    the x3 images)."""
    G = x2.shape[0]
    prec = int(X3 if x3 is None else x3)
    outs = [torch.empty(G, H, device=x2.device) for _ in blocks]
    if prec == 0:
        keep = [_image_f32(b_.t().contiguous() if transposed else b_) for b_ in blocks]
    else:
        keep = [_image(b_, 1, transposed=transposed, step=step, plan=plan) for b_ in blocks]          # x3 images serve codes 1 and 2
    bc = [None if b_ is None else b_.detach().contiguous() for b_ in biases]
    proj = (hip.NampProj * len(blocks))(*[hip.NampProj(keep[q].data_ptr(), hip.ptr(bc[q]), None, outs[q].data_ptr())
                                          for q in range(len(blocks))])
    hip.check(hip.lib().namp_node_linear_prec(x2.data_ptr(), G, proj, len(blocks), prec, hip.current_stream()), "node_linear")
    return outs


def _node_linear_sum(xs, blocks, x3=None, step=None, plan=True):
    """sum_q xs[q] @ blocks[q] over [G,128] rows in ONE launch (namp_node_linear_sum): the data gradient of _NodeLinears — as one node_linear launch per
    block the sum cost a stock addition per further block besides."""
    G = xs[0].shape[0]
    prec = int(X3 if x3 is None else x3)
    out = torch.empty(G, H, device=xs[0].device)
    if prec == 0:
        keep = [_image_f32(b_.t().contiguous()) for b_ in blocks]
    else:
        keep = [_image(b_, 1, transposed=True, step=step, plan=plan) for b_ in blocks]
    arr = lambda ptrs: (C.c_void_p * len(ptrs))(*ptrs)
    for q0 in range(0, len(xs), 8):                              # (more than 8 blocks: the launches' results are added)
        o = out if q0 == 0 else torch.empty_like(out)
        hip.check(hip.lib().namp_node_linear_sum(arr([x.data_ptr() for x in xs[q0:q0 + 8]]), arr([k.data_ptr() for k in keep[q0:q0 + 8]]),
                                                 len(xs[q0:q0 + 8]), o.data_ptr(), G, prec, hip.current_stream()), "node_linear_sum")
        if q0:
            out = out + o
    return out


class _NodeLinears(torch.autograd.Function):
    """Several residue-level linear maps of the same input, y_q = x W_q^T + b_q with W_q [128 x 128] column blocks of the layers'
    first-layer weights (the hoisted tables Pa / Pc / Pbw / Pfw, na_model_utils.py:218-236, 610-636, and W_v): one
    node_linear_kernel launch forward, one for dL/dx, the row-contraction kernel for dL/dW_q.  (The stock GEMM picks a
    32 x 32 tiling for these [24,000 x 128] x [128 x 128] products: 81 us each at cfg5.)"""

    @staticmethod
    def forward(ctx, x, nb, prec, *wb):
        Ws, bs = wb[:nb], wb[nb:]
        x2 = x.contiguous().view(-1, H)
        ctx.plan = prec is None                                 # an explicit precision comes with per-step temporaries (the padded output head)
        prec = X3 if prec is None else prec                     # (... and keeps the head fp32-equivalent under mixed precision)
        outs = _node_linear_call(x2, [w.detach() for w in Ws], bs, x3=prec, plan=ctx.plan)
        ctx.nb, ctx.shape, ctx.has_b, ctx.x3, ctx.step = nb, x.shape, [b is not None for b in bs], prec, _STEP
        ctx.save_for_backward(x2, *Ws)
        return tuple(o.view(*x.shape[:-1], H) for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        x2, *Ws = ctx.saved_tensors
        g2 = [g.contiguous().view(-1, H) for g in gs]
        gx = _node_linear_sum(g2, [w.detach() for w in Ws], x3=ctx.x3, step=ctx.step, plan=ctx.plan)      # dL/dx = sum_q g_q W_q
        res = _wgrad_many([(g2[q], x2, ctx.has_b[q]) for q in range(ctx.nb)], x3=ctx.x3)      # one reduction for all blocks
        gW, gb = [r[0] for r in res], [r[1] for r in res]
        return (gx.view(ctx.shape), None, None, *gW, *gb)


class _SplitCols(torch.autograd.Function):
    """W [128, n * 128] -> its n column blocks (views, as slicing gives them), with ONE concatenation as the backward.  Plain slices cost, per
    block and step, a zero fill of the full weight shape, a copy of the block's gradient into it and an in-place add to accumulate the blocks
    (8 stock launches per three-block weight; 33 fills, 34 copies, 24 adds per cfg5 step)."""

    @staticmethod
    def forward(ctx, W):
        ctx.n = W.shape[1] // H
        return tuple(W[:, q * H:(q + 1) * H] for q in range(ctx.n))

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        return torch.cat([g if g is not None else torch.zeros_like(ref) for g in gs], 1)


def _lin(x, *pairs):
    """_lin(x, (W_a, b_a), (W_b, None), ...) -> tuple of x W^T + b for [128 x 128] blocks; HIP kernels on a HIP device."""
    Ws = [p[0] for p in pairs]
    bs = [p[1] for p in pairs]
    return _NodeLinears.apply(x, len(Ws), None, *Ws, *bs)


class _RowLayerNorm(torch.autograd.Function):
    """LayerNorm over the 128 channels of an [..,128] tensor with E-sized leading dimensions (norm_edges on the edge embedding,
    na_model_utils.py:509): HIP forward / backward (ln_rows_*_kernel) instead of the stock op, which runs at ~1 TB/s on
    512-byte rows."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        out = torch.empty_like(x)
        rows = x.numel() // H
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        hip.check(hip.lib().namp_train_ln_rows_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), rows, hip.current_stream()),
                  "train_ln_rows_fwd")
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        rows = x.numel() // H
        gx = torch.empty_like(x)
        part = torch.empty(hip.lib().namp_train_ln_rows_groups(rows), 2, H, device=x.device)
        hip.check(hip.lib().namp_train_ln_rows_bwd(x.data_ptr(), g.data_ptr(), w.data_ptr(), gx.data_ptr(), part.data_ptr(), rows,
                                                   hip.current_stream()), "train_ln_rows_bwd")
        s = _reduce(_seg0(part))[0].view(2, H)
        return gx, s[0], s[1]


# dL/dy of the edge embedding as bf16 operand tiles (namp_train_embed_ln_bwd writes them beside the fp32 rows): keyed by the fp32 tensor's address so
# that the gradient launch of the embedding weight (_EdgeEmbeddingGrad.backward, the next autograd node) finds them IF it is handed that very tensor
_G16 = {}


class _EdgeEmbedTail(torch.autograd.Function):
    """h_E = W_e LayerNorm(y) + b_e (norm_edges + W_e, na_model_utils.py:509,598) with the normalised rows never in memory (split-bf16 / mixed
    precision): forward = one launch on the pre-LayerNorm rows (namp_edge_embed_ln); backward = one launch for dL/dy (W_e^T product + LayerNorm
    backward, namp_train_embed_ln_bwd) that also leaves the rows' (mean, rstd), and the row contraction dW_e = g^T LayerNorm(y) re-deriving its
    operand from y (namp_train_wgrad_ln).  Against _RowLayerNorm + _EdgeLinear: 1.2 GB less traffic forward, 1.2 GB less backward, one [E,128]
    tensor less alive from forward to backward."""

    @staticmethod
    def forward(ctx, y, ln_w, ln_b, W, b):
        B, N, K, _ = y.shape
        y = y.contiguous()
        out = torch.empty_like(y)
        lw, lb = ln_w.detach().contiguous(), ln_b.detach().contiguous()
        hip.check(hip.lib().namp_edge_embed_ln(_image(W.detach()).data_ptr(), b.detach().contiguous().data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                               y.data_ptr(), out.data_ptr(), int(X3), B, N, K, hip.current_stream()), "edge_embed_ln")
        ctx.x3, ctx.step = X3, _STEP
        ctx.save_for_backward(y, lw, lb, W)
        return out

    @staticmethod
    def backward(ctx, g):
        y, lw, lb, W = ctx.saved_tensors
        L = hip.lib()
        g = g.contiguous()
        rows = y.numel() // H
        dev = y.device
        g_pre = torch.empty_like(y)
        stats = torch.empty(rows, 2, device=dev)
        part = torch.empty(L.namp_train_embed_ln_bwd_groups(rows), 2, H, device=dev)
        # Mixed precision: dL/dy also as bf16 operand tiles for the embedding-weight gradient (feat_wgrad 2.17 -> 1.29 ms at cfg5 for +0.16 ms here).
        # Split-bf16 (hi + remainder tiles, G16_SPLIT): the gradient launch gains 0.05 ms and this one loses 0.35 — off.
        g16 = None
        if int(ctx.x3) == 2 or G16_SPLIT:
            ne = L.namp_train_g16_elems(rows)
            g16 = torch.empty(2 if int(ctx.x3) == 1 else 1, ne, device=dev, dtype=torch.bfloat16)      # [ceil(rows / 64)][128][64] (x 2)
            if rows % 64:
                g16[:, ne - 64 * H:].zero_()                  # rows past the end of the last tile
        hip.check(L.namp_train_embed_ln_bwd(g.data_ptr(), y.data_ptr(), _image_t(W, ctx.x3, ctx.step).data_ptr(), lw.data_ptr(), g_pre.data_ptr(),
                                            stats.data_ptr(), part.data_ptr(), hip.ptr(g16), int(ctx.x3), rows, hip.current_stream()),
                  "train_embed_ln_bwd")
        _G16.clear()
        if g16 is not None:
            _G16[g_pre.data_ptr()] = (g16, rows, int(ctx.x3), g_pre._version)
        n = L.namp_train_wgrad_chunks(rows)
        dWp, dbp = torch.empty(n, H, H, device=dev), torch.empty(n, H, device=dev)
        hip.check(L.namp_train_wgrad_ln(g.data_ptr(), y.data_ptr(), stats.data_ptr(), lw.data_ptr(), lb.data_ptr(), int(ctx.x3), rows,
                                        dWp.data_ptr(), dbp.data_ptr(), hip.current_stream()), "train_wgrad_ln")
        dgb, dW, db = _reduce(_seg0(part), _seg0(dWp), _seg0(dbp))
        dgb = dgb.view(2, H)
        return g_pre, dgb[0], dgb[1], dW.view(H, H), db.view(H)


def _ximage_general(W, transposed=False, step=None):
    """x3 image of a general [out_f x in_f] block (residue-level FFN weights; namp_pack_image_x3_general), cached per step."""
    step = _STEP if step is None else step
    use_cache = step is not None and step is _CACHE_STEP
    if use_cache:
        hit = _PLAN.lookup(W, 3, transposed, step)
        if hit is not None:
            return hit
    key = (W.data_ptr(), W.stride(0), W.stride(1), W._version, "xg", transposed, W.device.index)
    if use_cache and key in _IMG_CACHE:
        return _IMG_CACHE[key]
    if use_cache and W.dim() == 2 and W.stride(1) == 1 and W.dtype == torch.float32:
        _PLAN.register(W.detach(), 3, transposed)
    Wc = (W.detach().t() if transposed else W.detach()).contiguous().float()
    img = torch.empty(Wc.numel(), dtype=torch.float32, device=W.device)
    hip.check(hip.lib().namp_pack_image_x3_general(Wc.data_ptr(), Wc.shape[1], 0, Wc.shape[0], Wc.shape[1], img.data_ptr(),
                                                   hip.current_stream()), "pack_image_x3_general")
    if use_cache:
        _IMG_CACHE[key] = img
    return img


class _NodeTail(torch.autograd.Function):
    """The residue tail of EncLayer / DecLayer (na_model_utils.py:236-247, 268-283) as one HIP launch each way:
    out = mask * LayerNorm2(x1 + dropout2(W_out gelu(W_in x1 + b_in) + b_out)),  x1 = LayerNorm1(h_V + dropout1(dh)).
    Dropout masks are counter-based hashes regenerated in backward; x1, z and the pre-LayerNorm2 rows are kept (37 MB per layer at
    cfg5) instead of being recomputed; the four weight gradients go through the row-contraction kernel."""

    @staticmethod
    def forward(ctx, h_V, dh, mask32, ln1_w, ln1_b, W_in, b_in, W_out, b_out, ln2_w, ln2_b, p, seed1, seed2):
        L = hip.lib()
        shape = h_V.shape
        hv, d = h_V.contiguous().view(-1, H), dh.contiguous().view(-1, H)
        G = hv.shape[0]
        dev = hv.device
        out, x1, y = (torch.empty(G, H, device=dev) for _ in range(3))
        z = torch.empty(4, G, H, device=dev)
        c = lambda t: t.detach().contiguous()
        l1w, l1b, l2w, l2b, bi, bo = c(ln1_w), c(ln1_b), c(ln2_w), c(ln2_b), c(b_in), c(b_out)
        img_in, img_out = _ximage_general(W_in), _ximage_general(W_out)      # named: they must outlive the launch's enqueue
        hip.check(L.namp_train_tail_fwd(hv.data_ptr(), d.data_ptr(), hip.ptr(mask32), l1w.data_ptr(), l1b.data_ptr(),
                                        img_in.data_ptr(), bi.data_ptr(), img_out.data_ptr(),
                                        bo.data_ptr(), l2w.data_ptr(), l2b.data_ptr(), float(p), int(seed1), int(seed2),
                                        out.data_ptr(), x1.data_ptr(), z.data_ptr(), y.data_ptr(), G, hip.current_stream()),
                  "train_tail_fwd")
        ctx.p, ctx.seeds, ctx.shape, ctx.x3, ctx.step = float(p), (int(seed1), int(seed2)), shape, X3, _STEP
        ctx.save_for_backward(hv, d, mask32, ln1_w, ln2_w, W_in, W_out, x1, z, y)
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        hv, d, mask32, ln1_w, ln2_w, W_in, W_out, x1, z, y = ctx.saved_tensors
        L = hip.lib()
        G = hv.shape[0]
        dev = hv.device
        g = g.contiguous().view(G, H)
        g_hV, g_dh, g_f = (torch.empty(G, H, device=dev) for _ in range(3))
        g_z, h = torch.empty(4, G, H, device=dev), torch.empty(4, G, H, device=dev)
        part = torch.empty(L.namp_train_tail_groups(G), 4, H, device=dev)
        l1w, l2w = ln1_w.detach().contiguous(), ln2_w.detach().contiguous()
        img_outT, img_inT = _ximage_general(W_out, True, ctx.step), _ximage_general(W_in, True, ctx.step)
        hip.check(L.namp_train_tail_bwd(hv.data_ptr(), d.data_ptr(), hip.ptr(mask32), l1w.data_ptr(), l2w.data_ptr(),
                                        img_outT.data_ptr(), img_inT.data_ptr(),
                                        ctx.p, ctx.seeds[0], ctx.seeds[1], x1.data_ptr(), z.data_ptr(), y.data_ptr(), g.data_ptr(),
                                        g_hV.data_ptr(), g_dh.data_ptr(), g_f.data_ptr(), g_z.data_ptr(), h.data_ptr(), part.data_ptr(),
                                        G, hip.current_stream()), "train_tail_bwd")
        res = _wgrad_many([(g_f, h[q], q == 0) for q in range(4)] + [(g_z[q], x1, True) for q in range(4)], x3=ctx.x3)
        dW_out = torch.cat([res[q][0] for q in range(4)], 1)              # [128, 512]: column block q = g_f^T h_q
        db_out = res[0][1]
        dW_in = torch.cat([res[4 + q][0] for q in range(4)], 0)           # [512, 128]: row block q = g_z_q^T x1
        db_in = torch.cat([res[4 + q][1] for q in range(4)], 0)
        dln = _reduce(_seg0(part))[0].view(4, H)
        return (g_hV.view(ctx.shape), g_dh.view(ctx.shape), None, dln[2], dln[3], dW_in, db_in, dW_out, db_out, dln[0], dln[1],
                None, None, None)


def _tail(h_V, dh, mask32, maskf, p, drop, drop_p):
    """Residue tail of one layer: the fused HIP launches in the split-bf16 / mixed-precision modes, stock ops in exact fp32."""
    if X3 and h_V.is_cuda:
        s1, s2 = (torch.randint(0, 2 ** 31 - 1, (2,)).tolist() if drop_p > 0 else (0, 0))      # host RNG: follows torch.manual_seed
        return _NodeTail.apply(h_V, dh, mask32, p.norm1.weight, p.norm1.bias, p.dense.W_in.weight, p.dense.W_in.bias,
                               p.dense.W_out.weight, p.dense.W_out.bias, p.norm2.weight, p.norm2.bias, drop_p, s1, s2)
    h_V = _ln(h_V + drop(dh), p.norm1)
    return maskf * _ln(h_V + drop(_ffn(h_V, p.dense)), p.norm2)


def _ffn(x, dense):
    """PositionWiseFeedForward (na_model_utils.py:286-296) on stock ops: only the exact-fp32 mode comes here — in the
    split-bf16 AND the mixed-precision (bf16) modes `_tail` takes the fused `_NodeTail` launches, whose GEMMs are split-bf16
    products in both (the residue-level math stays fp32-equivalent under mixed precision; only the per-edge GEMMs are plain bf16)."""
    return dense.W_out(F.gelu(dense.W_in(x)))


def forward_train(model, fd, decoding_randn=None):
    """Differentiable ProteinMPNN.forward of the training copy (na_model_utils.py:589-646) -> (log_probs, probs)."""
    global X3, _STEP, _CACHE_STEP
    X3 = PREC_CODE[getattr(model, "message_precision", "x3")]
    _STEP = _CACHE_STEP = object()                           # a new step: its own (empty) image cache
    _IMG_CACHE.clear()
    _G16.clear()
    if X3 and fd["mask"].is_cuda and torch.is_grad_enabled():
        _PLAN.begin_step(_STEP, fd["mask"].device)            # every image the previous step asked for, from one launch
    try:
        return _forward_train(model, fd, decoding_randn)
    finally:
        _STEP = None


def _forward_train(model, fd, decoding_randn):
    mask = fd["mask"]
    if not mask.is_cuda:
        raise RuntimeError("na_mpnn_amd.train: tensors must be on a HIP device (no CPU fallback)")
    drop_p = float(model.dropout.p) if model.training else 0.0
    drop = (lambda t: F.dropout(t, drop_p, True)) if drop_p > 0 else (lambda t: t)
    fp = model.features
    y, E_idx = edge_embedding(model, fd)
    B, N, K = E_idx.shape
    V = _ln(_TableRows.apply(fp.node_embedding.weight.t(), fd["R_polymer_type"].long()), fp.norm_nodes)   # one-hot @ W^T
    h_V = _lin(V, (model.W_v.weight, model.W_v.bias))[0]
    if X3 and EMBED_LN_FUSED:
        h_E = _EdgeEmbedTail.apply(y, fp.norm_edges.weight, fp.norm_edges.bias, model.W_e.weight, model.W_e.bias)
    else:
        h_E = _EdgeLinear.apply(_RowLayerNorm.apply(y, fp.norm_edges.weight, fp.norm_edges.bias), model.W_e.weight, model.W_e.bias)
    mask32 = mask.to(torch.int32).contiguous()
    maskf = mask.float().unsqueeze(-1)
    rev = ReverseAdjacency(E_idx) if torch.is_grad_enabled() else None
    for p in model.encoder_layers:                                                   # EncLayer, na_model_utils.py:218-241
        W1a, W1b, W1c = _SplitCols.apply(p.W1.weight)
        W11a, W11b, W11c = _SplitCols.apply(p.W11.weight)
        Pa, Pc = _lin(h_V, (W1a, p.W1.bias), (W1c, None))
        dh, h_E = _EdgeMLP.apply(ENC_MSG, h_E, Pa, Pc, None, W1b, p.W2.weight, p.W2.bias, p.W3.weight, p.W3.bias,
                                 E_idx, mask32, None, None, rev)           # h_E: passed through to the edge update below
        h_V = _tail(h_V, dh, mask32, maskf, p, drop, drop_p)
        Pa, Pc = _lin(h_V, (W11a, p.W11.bias), (W11c, None))
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop_p > 0 else 0      # host RNG: follows torch.manual_seed
        h_E = _EdgeUpdate.apply(h_E, Pa, Pc, W11b, p.W12.weight, p.W12.bias, p.W13.weight, p.W13.bias,
                                p.norm3.weight, p.norm3.bias, E_idx, drop_p, seed, rev)
    chain_M = mask
    if model.decode_protein_first:
        chain_M = chain_M.masked_fill(fd["protein_mask"].to(torch.bool), 0.0)
    if decoding_randn is None:
        decoding_randn = torch.randn(chain_M.shape, device=mask.device)
    rank32 = model.ranks_of(model.decoding_order(chain_M, decoding_randn)).to(torch.int32).contiguous()
    h_S = _TableRows.apply(model.W_s.weight, fd["S"].long())
    h_V_enc = h_V
    for p in model.decoder_layers:                                                   # DecLayer on the implicit h_ESV, :610-640
        W1a, W1e, W1s, W1v = _SplitCols.apply(p.W1.weight)
        Pa, Pv = _lin(h_V, (W1a, p.W1.bias), (W1v, None))
        Pbw = _lin(h_S, (W1s, None))[0] + Pv
        Pfw = _lin(h_V_enc, (W1v, None))[0]
        dh, h_E = _EdgeMLP.apply(DEC_MSG, h_E, Pa, Pbw, Pfw, W1e, p.W2.weight, p.W2.bias, p.W3.weight, p.W3.bias,
                                 E_idx, None, None, rank32, rev)          # h_E: passed through to the next DecLayer
        h_V = _tail(h_V, dh, mask32, maskf, p, drop, drop_p)
    # output head (33 rows) on the residue-level HIP launches: W_out padded to a [128 x 128] block, the extra 95 logits dropped
    nl = model.W_out.weight.shape[0]
    logits = _NodeLinears.apply(h_V, 1, min(int(X3), 1), F.pad(model.W_out.weight, (0, 0, 0, H - nl)),
                                F.pad(model.W_out.bias, (0, H - nl)))[0][..., :nl]
    return F.log_softmax(logits, dim=-1), F.softmax(logits, dim=-1)


# ------------------------------------------------------------------------------------------------------------
# loss / optimiser / checkpoints  (na_model_utils.py:100-146, 648-686; na_run.py:100-130, 330-353)
# ------------------------------------------------------------------------------------------------------------
def loss_nll(S, log_probs, mask):
    """na_model_utils.py:100-109."""
    loss = F.nll_loss(log_probs.contiguous().view(-1, log_probs.size(-1)), S.contiguous().view(-1), reduction="none").view(S.size())
    true_false = (S == torch.argmax(log_probs, -1)).float()
    return loss, torch.sum(loss * mask) / torch.sum(mask), true_false


class _LossSmoothed(torch.autograd.Function):
    """Per-residue label-smoothed loss (fp64) on the HIP kernel loss_smoothed_kernel, one launch each way."""

    @staticmethod
    def forward(ctx, log_probs, S32, pm, rm, eps3, weight, ppm_mask32, aligned_ppm64):
        G, V = log_probs.numel() // log_probs.shape[-1], log_probs.shape[-1]
        lp = log_probs.contiguous().float()
        loss = torch.empty(log_probs.shape[:-1], dtype=torch.float64, device=lp.device)
        ctx.args = (S32, pm, rm, eps3, float(weight), ppm_mask32, aligned_ppm64, G, V)
        ctx.lp_shape = log_probs.shape
        hip.check(hip.lib().namp_train_loss_smoothed(0, S32.data_ptr(), lp.data_ptr(), *[t.data_ptr() for t in pm], *[t.data_ptr() for t in rm],
                                                     eps3, float(weight), hip.ptr(ppm_mask32), hip.ptr(aligned_ppm64), loss.data_ptr(), None, None,
                                                     G, V, hip.current_stream()), "train_loss_smoothed")
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        S32, pm, rm, eps3, weight, ppm_mask32, aligned_ppm64, G, V = ctx.args
        g = g_loss.contiguous().double()
        g_lp = torch.empty(ctx.lp_shape, dtype=torch.float32, device=g.device)
        hip.check(hip.lib().namp_train_loss_smoothed(1, S32.data_ptr(), None, *[t.data_ptr() for t in pm], *[t.data_ptr() for t in rm],
                                                     eps3, weight, hip.ptr(ppm_mask32), hip.ptr(aligned_ppm64), None, g.data_ptr(), g_lp.data_ptr(),
                                                     G, V, hip.current_stream()), "train_loss_smoothed_bwd")
        return g_lp, None, None, None, None, None, None, None


def loss_smoothed(S, log_probs, mask, polymer_masks, polymer_restype_masks, polymer_restype_nums, weight=0.1, tokens=2000.0,
                  num_letters=33, ppm_mask=None, aligned_ppm=None):
    """Label-smoothed negative log-likelihood in fp64 with per-polymer smoothing mass (na_model_utils.py:111-146).  On a HIP device
    the per-residue loss and its gradient are one launch each (`_LossSmoothed`); host tensors take the stock-op restatement below."""
    if log_probs.is_cuda:
        keys = ("protein", "dna", "rna")
        f32 = lambda t: t.contiguous().to(torch.float32)
        pm = [f32(polymer_masks[k]) for k in keys]
        rm = [f32(polymer_restype_masks[k]) for k in keys]
        eps3 = (C.c_float * 3)(*[float(np.float32(weight / polymer_restype_nums[k])) for k in keys])
        S32 = S.contiguous().to(torch.int32)
        pm32 = ppm_mask.contiguous().to(torch.int32) if ppm_mask is not None else None
        ppm64 = aligned_ppm.contiguous().to(torch.float64) if ppm_mask is not None else None
        loss = _LossSmoothed.apply(log_probs, S32, pm, rm, eps3, weight, pm32, ppm64)
        return loss, torch.sum(loss * mask) / tokens
    pr, dr, rr = (polymer_restype_masks[k] for k in ("protein", "dna", "rna"))
    onehot = F.one_hot(S, num_letters).to(torch.float64)
    if ppm_mask is not None:
        onehot[ppm_mask.bool()] = aligned_ppm[ppm_mask.bool()]
    eps = sum(polymer_masks[k][:, :, None] * polymer_restype_masks[k][None, None, :] * (weight / polymer_restype_nums[k])
              for k in ("protein", "dna", "rna"))
    onehot[:, :, (pr + dr + rr).bool()] *= (1 - weight)
    onehot = onehot + eps
    loss = -(onehot * log_probs).sum(-1)
    return loss, torch.sum(loss * mask) / tokens


class NoamOpt:
    """Learning-rate schedule wrapper (na_model_utils.py:648-680): lr = factor * d^-0.5 * min(step^-0.5, step * warmup^-1.5)."""

    def __init__(self, model_size, factor, warmup, optimizer, step):
        self.optimizer, self._step, self.warmup, self.factor, self.model_size, self._rate = \
            optimizer, step, warmup, factor, model_size, 0

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def rate(self, step=None):
        step = self._step if step is None else step
        return self.factor * (self.model_size ** (-0.5) * min(step ** (-0.5), step * self.warmup ** (-1.5)))

    def step(self):
        self._step += 1
        self._rate = self.rate()
        for p in self.optimizer.param_groups:
            p["lr"] = self._rate
        self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam whose step — optionally preceded by torch.nn.utils.clip_grad_norm_ (`clip_norm`, na_run.py:233-236) — runs as
    ONE multi-tensor HIP launch over all parameter tensors (namp_train_adam_step; three launches with clipping) instead of the
    foreach kernel chain.  The state is torch's own (`step` / `exp_avg` / `exp_avg_sq` per parameter), so `state_dict()` is the
    reference's checkpoint format (na_run.py:342) and loads into a plain torch.optim.Adam.  Plain Adam only (no amsgrad / weight decay
    / maximize), fp32 parameters on one HIP device; anything else falls back to torch's step.  The step count is read from the state's `step`
    tensors once per set of tensors and counted on the host afterwards: REPLACE those tensors to change it (as load_state_dict does) — an
    in-place edit of state["step"] is not seen until they are replaced."""

    clip_norm = 0.0            # > 0: clip the global gradient norm to this value inside the step
    last_grad_norm = None      # device tensor [2]: (gradient norm, clip coefficient) of the last clipped step

    def _fused_ok(self, params):
        g0 = self.param_groups[0]
        return (len(self.param_groups) == 1 and not g0.get("amsgrad") and g0.get("weight_decay", 0) == 0 and not g0.get("maximize")
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32 and not p.grad.is_sparse for p in params))

    @torch.no_grad()
    def step(self, closure=None):
        params = [p for p in self.param_groups[0]["params"] if p.grad is not None] if len(self.param_groups) == 1 else []
        if closure is not None or not params or not self._fused_ok(params):
            if self.clip_norm > 0:
                torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"]], self.clip_norm)
            return super().step(closure)
        grp = self.param_groups[0]
        dev = params[0].device
        for p in params:                                      # torch's lazy state initialisation (_init_group)
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        steps = [self.state[p]["step"] for p in params]
        # the step counts are read (and checked equal) ONCE per set of state tensors — at the first step and after load_state_dict —
        # and counted on the host afterwards: with capturable / device-side state every .item() would be a device synchronisation
        skey = tuple(id(s_) for s_ in steps)
        if getattr(self, "_t_key", None) != skey:
            t0 = int(steps[0].item())
            if any(int(s_.item()) != t0 for s_ in steps[1:]):
                raise RuntimeError("FusedAdam: parameters with different step counts (load a consistent optimizer_state_dict)")
            self._t, self._t_key = t0, skey
        torch._foreach_add_(steps, 1)
        self._t += 1
        t = self._t
        key = tuple((p.data_ptr(), p.numel()) for p in params)
        if getattr(self, "_plan_key", None) != key:
            chunk = hip.lib().namp_train_adam_chunk()
            bt, bo = [], []
            for i, p in enumerate(params):
                for off in range(0, p.numel(), chunk):
                    bt.append(i); bo.append(off)
            self._plan = (torch.tensor(bt, dtype=torch.int32, device=dev), torch.tensor(bo, dtype=torch.int64, device=dev),
                          torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev), len(bt))
            self._ws = torch.empty(len(bt) + 2, dtype=torch.float32, device=dev)
            self._plan_key = key
        bt, bo, numel, nb = self._plan
        # The 4 x ntensors pointer table: cached on the device while the parameter / gradient / state addresses are unchanged (the caching
        # allocator hands the same gradient blocks back step after step), otherwise uploaded from a PINNED staging buffer without blocking —
        # a pageable .to(dev) waits for the whole backward pass on the stream and leaves the device idle until the host has restarted
        # (measured: ~0.5 ms of every cfg5 step)
        rows = ([p.data_ptr() for p in params], [p.grad.data_ptr() for p in params],
                [self.state[p]["exp_avg"].data_ptr() for p in params], [self.state[p]["exp_avg_sq"].data_ptr() for p in params])
        pkey = tuple(map(tuple, rows))
        if getattr(self, "_ptr_key", None) != pkey:
            ring = getattr(self, "_ptr_ring", None)
            if ring is None or ring[0].shape[1] != len(params):
                ring = self._ptr_ring = [torch.empty(4, len(params), dtype=torch.int64).pin_memory() for _ in range(4)]
                self._ptr_dev = [torch.empty(4, len(params), dtype=torch.int64, device=dev) for _ in range(4)]
                self._ptr_i = 0
            self._ptr_i = (self._ptr_i + 1) % 4            # (four staging buffers: a buffer is rewritten four uploads later at the earliest ...
            ev = getattr(self, "_ptr_ev", None)
            if ev is None or len(ev) != 4:
                ev = self._ptr_ev = [None] * 4
            if ev[self._ptr_i] is not None:
                ev[self._ptr_i].synchronize()              # ... and never before the upload that last read it has completed)
            ring[self._ptr_i].copy_(torch.tensor(rows, dtype=torch.int64))
            self._ptr_dev[self._ptr_i].copy_(ring[self._ptr_i], non_blocking=True)
            ev[self._ptr_i] = torch.cuda.Event()
            ev[self._ptr_i].record()
            self._ptrs, self._ptr_key = self._ptr_dev[self._ptr_i], pkey
        ptrs = self._ptrs
        b1, b2 = grp["betas"]
        lr = float(grp["lr"])
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        hip.check(hip.lib().namp_train_adam_step(bt.data_ptr(), bo.data_ptr(), numel.data_ptr(), ptrs.data_ptr(), len(params), nb,
                                                 float(self.clip_norm), float(b1), float(b2), lr / bc1, bc2 ** 0.5, float(grp["eps"]),
                                                 self._ws.data_ptr(), hip.current_stream()), "train_adam_step")
        self.last_grad_norm = self._ws[:2] if self.clip_norm > 0 else None
        return None


def get_std_opt(parameters, d_model, step):
    """na_model_utils.py:682-686 (Adam lr 0, betas (0.9, 0.98), eps 1e-9 under the Noam schedule) on the multi-tensor HIP step."""
    return NoamOpt(d_model, 2, 4000, FusedAdam(parameters, lr=0, betas=(0.9, 0.98), eps=1e-9), step)


def save_checkpoint(path, model, optimizer, epoch, step, save_step=None):
    """The reference's checkpoint dict (na_run.py:330-353), loadable by inference/run.py:198-200."""
    torch.save({"epoch": epoch, "step": step, "save_step": step if save_step is None else save_step,
                "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.optimizer.state_dict()}, path)


def load_checkpoint(path, model, optimizer=None, map_location=None):
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.optimizer.load_state_dict(ck["optimizer_state_dict"])
        optimizer._step = ck.get("step", optimizer._step)
    return ck


def polymer_restype_tables(restype_to_int, num_letters, device):
    """0/1 vectors over the vocabulary marking each polymer's residue types and the list lengths the reference
    divides the smoothing mass by (na_data_utils.py:185-223,281-283; na_run.py:138-154): 21 / 5 / 5."""
    names = {"protein": spec.RESTYPES[:20] + ["UNK"], "dna": ["DA", "DC", "DG", "DT", "DX"], "rna": ["A", "C", "G", "U", "RX"]}
    masks, nums = {}, {}
    for key, lst in names.items():
        v = torch.zeros(num_letters, device=device)
        v[[restype_to_int[n] for n in lst]] = 1
        masks[key], nums[key] = v, len(lst)
    return masks, nums


def train_step(model, optimizer, fd, polymer_restype_masks, polymer_restype_nums, tokens_with_no_loss, label_smoothing=0.1,
               loss_tokens=2000.0, gradient_norm=0.0, decoding_randn=None, data_parallel=False, scaler=None):
    """One optimisation step of na_run.py:198-238: forward, label-smoothed loss, backward, clip, Noam/Adam.
    Mixed precision (the reference's MIXED_PRECISION branch, :216-238): set ``model.message_precision = "bf16"`` — the
    per-edge GEMMs of forward, backward and weight gradients then run as plain bf16 products with fp32 accumulation, master
    weights / residue-level math / loss / optimiser stay fp32 — and optionally pass the reference's ``scaler``
    (torch GradScaler): ``scaler.scale(loss).backward(); clip; scaler.step(optimizer); scaler.update()`` like :232-238.
    ``data_parallel`` (an extension; torch.distributed initialised, one process per GPU): gradients are averaged over
    the ranks with one all-reduce before clipping, so every rank applies the same update."""
    optimizer.zero_grad()
    S, mask = fd["S"].long(), fd["mask"]
    S_mask = 1 - torch.any(S[:, :, None] == tokens_with_no_loss[None, None, :], dim=-1).long()
    mask_for_loss = mask * S_mask
    polymer_masks = {"protein": fd["protein_mask"], "dna": fd["dna_mask"], "rna": fd["rna_mask"]}
    log_probs, _ = forward_train(model, fd, decoding_randn)
    # position-probability targets of the specificity model (na_run.py:229-230), when the batch carries them
    _, loss = loss_smoothed(S, log_probs, mask_for_loss, polymer_masks, polymer_restype_masks, polymer_restype_nums,
                            weight=label_smoothing, tokens=loss_tokens, num_letters=log_probs.shape[-1],
                            ppm_mask=fd.get("ppm_mask"), aligned_ppm=fd.get("aligned_ppm"))
    (scaler.scale(loss) if scaler is not None else loss).backward()
    if data_parallel:
        from . import shard
        shard.allreduce_gradients(model.parameters())          # one RCCL all-reduce of the 9.2 MB gradient bucket
    fused = isinstance(getattr(optimizer, "optimizer", None), FusedAdam) and scaler is None
    if fused:
        optimizer.optimizer.clip_norm = float(gradient_norm)       # clip + Noam rate + Adam: one multi-tensor launch sequence
    elif gradient_norm > 0.0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_norm)
    if scaler is not None:
        if isinstance(getattr(optimizer, "optimizer", None), FusedAdam):
            optimizer.optimizer.clip_norm = 0.0
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    return loss.detach(), log_probs.detach()
