"""CPU emulation of the product's MIXED-PRECISION training step (`model.message_precision = "bf16"`) — TEST INFRASTRUCTURE ONLY.

The reference trains under `torch.cuda.amp.autocast` + `GradScaler` (na_run.py:21,216-238); autocast's bf16 rounding points on
the CPU are not the HIP kernels', so the product's mixed mode is checked against THIS restatement: the reference's training
forward (na_model_utils.py:589-646, as restated in oracle/cpu_ref.py) with exactly the per-edge GEMMs evaluated the way the kernels
evaluate them in that mode — operands rounded to bf16 (round-to-nearest-even), products and sums in fp32 — in the forward pass, in the
data-gradient GEMMs (the upstream gradient is rounded where it enters a product) and in the weight-gradient contractions (both row
operands rounded); everything else (residue-level linear layers, LayerNorms, K-sums, loss) in fp32 like cpu_ref.  The per-edge GEMMs are
  * the 5200 -> 128 edge embedding on [positional | RBF] features      (edge_features_kernel<2>, feat_wgrad_x3_kernel<false>)
  * W_e                                                               (edge_mlp_kernel<MODE_EMBED> bf16, wgrad_x3_kernel<false>)
  * W1b / W2 of every message MLP; layer 3 acts on the K-sum, per residue, in fp32   
  * W11b / W12 / W13 of every edge update.
The hoisted first-layer terms W1a.h_V_i + b1 and W1c.h_V_j are residue-level fp32 products that enter z1 as additions; the gradient
flowing back into the gathered table passes through the bf16 row tensor G1 (`namp_train_scatter_rows_bf16`), the one flowing into
Pa[i] does not (summed from the fp32 registers).  GELU is exact-erf here; the kernels' forward uses a polynomial of 1.9e-4.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import cpu_ref as R


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _Bf16Linear(torch.autograd.Function):
    """y = bf16(x) . bf16(W)^T (+ b) with fp32 accumulation; dx = bf16(g) . bf16(W); dW = bf16(g)^T . bf16(x); db = sum g (fp32)."""

    @staticmethod
    def forward(ctx, x, W, b):
        xr, Wr = bf16r(x), bf16r(W)
        ctx.save_for_backward(xr, Wr)
        ctx.has_b = b is not None
        y = xr @ Wr.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, g):
        xr, Wr = ctx.saved_tensors
        gr = bf16r(g)
        g2, x2 = gr.reshape(-1, gr.shape[-1]), xr.reshape(-1, xr.shape[-1])
        return gr @ Wr, g2.t() @ x2, (g.reshape(-1, g.shape[-1]).sum(0) if ctx.has_b else None)


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded to bf16 (a gradient that reaches its consumer through a stored bf16 row tensor)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf16r(g)


def blin(x, W, b=None):
    return _Bf16Linear.apply(x, W, b)


def _edge_mlp_pre(w, p, names, h_V, h_E, E_idx):
    """gelu(W2 . gelu(z1) + b2) with z1 = W1b.h_E (bf16 product) + (W1a.h_V_i + b1) + W1c.h_V_j (fp32 tables)."""
    W1, b1 = w[p + names[0] + ".weight"], w[p + names[0] + ".bias"]
    H = R.H
    Pa = F.linear(h_V, W1[:, :H], b1)
    Pj = F.linear(h_V, W1[:, 2 * H:3 * H])
    z1 = blin(h_E, W1[:, H:2 * H]) + Pa.unsqueeze(-2) + _RoundGrad.apply(R.gather_nodes(Pj, E_idx))
    return F.gelu(blin(F.gelu(z1), w[p + names[1] + ".weight"], w[p + names[1] + ".bias"]))


def enc_layer(w, p, h_V, h_E, E_idx, mask_V, mask_attend):
    a2 = _edge_mlp_pre(w, p, ("W1", "W2", "W3"), h_V, h_E, E_idx)
    wgt = mask_attend.unsqueeze(-1) / R.SCALE
    S, ws = (wgt * a2).sum(-2), wgt.sum(-2)
    dh = F.linear(S, w[p + "W3.weight"]) + ws * w[p + "W3.bias"]                   # layer 3 behind the K-sum, fp32, per residue
    h_V = R._ln(w, p + "norm1", h_V + dh)
    h_V = R._ln(w, p + "norm2", h_V + R.ffn(w, p + "dense.", h_V))
    h_V = mask_V.unsqueeze(-1) * h_V
    a2 = _edge_mlp_pre(w, p, ("W11", "W12", "W13"), h_V, h_E, E_idx)
    h_E = R._ln(w, p + "norm3", h_E + blin(a2, w[p + "W13.weight"], w[p + "W13.bias"]))
    return h_V, h_E


def dec_layer(w, p, h_V, h_E, h_S, h_V_enc, E_idx, mask, bw):
    """DecLayer on the implicit context [h_V_i | mask_i h_E | mask_bw W_s[S_j] | mask_bw h_V_j + mask_fw h_V^enc_j]."""
    H = R.H
    W1, b1 = w[p + "W1.weight"], w[p + "W1.bias"]
    m1 = mask.view(mask.shape[0], mask.shape[1], 1, 1).float()
    Pa = F.linear(h_V, W1[:, :H], b1)
    Pbw = F.linear(h_S, W1[:, 2 * H:3 * H]) + F.linear(h_V, W1[:, 3 * H:])
    Pfw = F.linear(h_V_enc, W1[:, 3 * H:])
    Pj = bw * R.gather_nodes(Pbw, E_idx) + (1. - bw) * R.gather_nodes(Pfw, E_idx)
    z1 = blin(m1 * h_E, W1[:, H:2 * H]) + Pa.unsqueeze(-2) + _RoundGrad.apply(m1 * Pj)
    a2 = F.gelu(blin(F.gelu(z1), w[p + "W2.weight"], w[p + "W2.bias"]))
    S = a2.sum(-2) / R.SCALE
    dh = F.linear(S, w[p + "W3.weight"]) + (a2.shape[-2] / R.SCALE) * w[p + "W3.bias"]
    h_V = R._ln(w, p + "norm1", h_V + dh)
    h_V = R._ln(w, p + "norm2", h_V + R.ffn(w, p + "dense.", h_V))
    return mask.unsqueeze(-1) * h_V


def features(w, fd, top_k):
    """cpu_ref.features with the edge embedding as a bf16 product."""
    X, mask = fd["X"], fd["mask"]
    A = R.ATOM
    Ca, N, C = X[:, :, A["CA"], :], X[:, :, A["N"], :], X[:, :, A["C"], :]
    Cb = R._virtual_atom(N, Ca, C, -0.58273431, 0.56802827, -0.54067466)
    N_na = R._virtual_atom(X[:, :, A["O4'"], :], X[:, :, A["C1'"], :], X[:, :, A["C2'"], :], -0.56967352, 0.51055973, -0.53122153)
    X18 = torch.cat((X, Cb[:, :, None, :], N_na[:, :, None, :]), -2)
    M18 = torch.cat((fd["X_m"], fd["protein_mask"][:, :, None], (fd["rna_mask"] + fd["dna_mask"])[:, :, None]), -1)
    _, E_idx = R.knn(Ca + X[:, :, A["C1'"], :], mask, top_k)
    Rb = R.rbf_all_pairs(X18, E_idx, M18)
    R_idx, chain = fd["R_idx"], fd["chain_labels"]
    offset = R.gather_edges((R_idx[:, :, None] - R_idx[:, None, :])[:, :, :, None], E_idx)[:, :, :, 0]
    same = R.gather_edges(((chain[:, :, None] - chain[:, None, :]) == 0).long()[:, :, :, None], E_idx)[:, :, :, 0]
    feat = torch.cat((R.positional(w, offset.long(), same), Rb), -1)
    E = R._ln(w, "features.norm_edges", blin(feat, w["features.edge_embedding.weight"]))
    V = F.one_hot(fd["R_polymer_type"], num_classes=w["features.node_embedding.weight"].shape[1]).float()
    V = R._ln(w, "features.norm_nodes", R._lin(w, "features.node_embedding", V))
    return V, E, E_idx


def forward_train(w, fd, top_k, randn):
    V, E, E_idx = features(w, fd, top_k)
    mask = fd["mask"]
    h_V = R._lin(w, "W_v", V)
    h_E = blin(E, w["W_e.weight"], w["W_e.bias"])
    mask_attend = mask.unsqueeze(-1) * R.gather_nodes(mask.unsqueeze(-1), E_idx).squeeze(-1)
    for i in range(R.n_layers(w, "encoder")):
        h_V, h_E = enc_layer(w, f"encoder_layers.{i}.", h_V, h_E, E_idx, mask, mask_attend)
    order = R.decoding_order_of(mask, randn)
    bw = R.backward_mask(order, E_idx)
    h_S = F.embedding(fd["S"].long(), w["W_s.weight"])
    h_V_enc = h_V
    for i in range(R.n_layers(w, "decoder")):
        h_V = dec_layer(w, f"decoder_layers.{i}.", h_V, h_E, h_S, h_V_enc, E_idx, mask, bw)
    logits = R._lin(w, "W_out", h_V)
    return F.log_softmax(logits, dim=-1), F.softmax(logits, dim=-1)


def train_loss_and_grads(w, fd, top_k, randn, restype_to_int, weight=0.1, tokens=2000.0):
    """cpu_ref.train_loss_and_grads on the mixed-precision emulation: loss (fp64), log_probs, {key: grad}."""
    wg = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    with torch.enable_grad():
        log_probs, _ = forward_train(wg, fd, top_k, randn)
        S = fd["S"].long()
        no_loss = torch.tensor([restype_to_int[t] for t in R.NO_LOSS_TOKENS])
        S_mask = 1 - torch.any(S[:, :, None] == no_loss[None, None, :], dim=-1).long()
        rm, rn = R.restype_masks(restype_to_int, log_probs.shape[-1])
        pm = {"protein": fd["protein_mask"], "dna": fd["dna_mask"], "rna": fd["rna_mask"]}
        _, loss = R.loss_smoothed(S, log_probs, fd["mask"] * S_mask, pm, rm, rn, weight, tokens, log_probs.shape[-1])
        loss.backward()
    return loss.detach(), log_probs.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in wg.items()}


def forward_train_exact(w, fd, top_k, randn):
    """The same hoisted formulation with exact fp32 products (bf16r -> identity): must reproduce cpu_ref.forward_train to fp32
    round-off — the check that the restructuring above (hoisted tables, layer 3 behind the K-sum, implicit decoder context) is
    the reference's function."""
    global bf16r
    keep = bf16r
    bf16r = lambda x: x
    try:
        return forward_train(w, fd, top_k, randn)
    finally:
        bf16r = keep
