"""CPU ORACLE — test infrastructure, never the product path.

An eager-PyTorch-CPU *restatement* (functional, weights passed as a flat
``{state_dict key: tensor}`` mapping) of the NA-MPNN message-passing hot path.
Each function cites the reference lines it follows.  It performs the same ATen
op sequence as the reference — including the materialised neighbour
concatenations — so that (i) on CPU it is bit-identical to the reference (pinned
by ``tests/golden/*.npz``, produced by ``oracle/make_goldens.py`` which imports
/root/reference in the build container), and (ii) it has the reference's cost
profile when ``bench.py`` times it as ``cpu_baseline`` (kind "port").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this file.  ``na_mpnn_amd`` itself must never do so.

Parity status: PINNED by import-based goldens (the reference has no tests of its
own — SURVEY §4/§8(c)).
"""
from __future__ import annotations

import itertools

import numpy as np
import torch
import torch.nn.functional as F

H = 128
SCALE = 30.0


# ----------------------------------------------------------------------------------------
# a1-a3: neighbour indexing (inference/model_utils.py:707-732, na_model_utils.py:168-193)
# ----------------------------------------------------------------------------------------
def gather_nodes(nodes, idx):
    """nodes [B,N,C], idx [B,N,K] -> [B,N,K,C]   (model_utils.py:713-721)."""
    b, n, k = idx.shape
    flat = idx.reshape(b, n * k).unsqueeze(-1).expand(-1, -1, nodes.size(2))
    return torch.gather(nodes, 1, flat).view(b, n, k, -1)


def gather_edges(edges, idx):
    """edges [B,N,N,C], idx [B,N,K] -> [B,N,K,C]   (model_utils.py:707-711)."""
    return torch.gather(edges, 2, idx.unsqueeze(-1).expand(-1, -1, -1, edges.size(-1)))


def cat_neighbors_nodes(h_nodes, h_neighbors, idx):
    """[h_neighbors | h_nodes[idx]] along channels   (model_utils.py:729-732)."""
    return torch.cat([h_neighbors, gather_nodes(h_nodes, idx)], -1)


# ----------------------------------------------------------------------------------------
# a4-a6: layers
# ----------------------------------------------------------------------------------------
def _lin(w, name, x):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def _ln(w, name, x):
    return F.layer_norm(x, (x.size(-1),), w[name + ".weight"], w[name + ".bias"], 1e-5)


def ffn(w, p, x):
    """PositionWiseFeedForward (model_utils.py:595-604)."""
    return _lin(w, p + "W_out", F.gelu(_lin(w, p + "W_in", x)))


def _mlp3(w, p, names, x):
    a, b, c = names
    return _lin(w, p + c, F.gelu(_lin(w, p + b, F.gelu(_lin(w, p + a, x)))))


def enc_layer(w, p, h_V, h_E, E_idx, mask_V=None, mask_attend=None):
    """EncLayer.forward, dropout inactive (model_utils.py:681-704)."""
    k = E_idx.size(-1)
    h_EV = cat_neighbors_nodes(h_V, h_E, E_idx)
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, k, -1), h_EV], -1)
    msg = _mlp3(w, p, ("W1", "W2", "W3"), h_EV)
    if mask_attend is not None:
        msg = mask_attend.unsqueeze(-1) * msg
    dh = torch.sum(msg, -2) / SCALE
    h_V = _ln(w, p + "norm1", h_V + dh)
    h_V = _ln(w, p + "norm2", h_V + ffn(w, p + "dense.", h_V))
    if mask_V is not None:
        h_V = mask_V.unsqueeze(-1) * h_V
    h_EV = cat_neighbors_nodes(h_V, h_E, E_idx)
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, k, -1), h_EV], -1)
    h_E = _ln(w, p + "norm3", h_E + _mlp3(w, p, ("W11", "W12", "W13"), h_EV))
    return h_V, h_E


def dec_layer(w, p, h_V, h_E, mask_V=None, mask_attend=None):
    """DecLayer.forward, dropout inactive (model_utils.py:636-657)."""
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, h_E.size(-2), -1), h_E], -1)
    msg = _mlp3(w, p, ("W1", "W2", "W3"), h_EV)
    if mask_attend is not None:
        msg = mask_attend.unsqueeze(-1) * msg
    dh = torch.sum(msg, -2) / SCALE
    h_V = _ln(w, p + "norm1", h_V + dh)
    h_V = _ln(w, p + "norm2", h_V + ffn(w, p + "dense.", h_V))
    if mask_V is not None:
        h_V = mask_V.unsqueeze(-1) * h_V
    return h_V


def n_layers(w, kind):
    return 1 + max(int(k.split(".")[1]) for k in w if k.startswith(kind + "_layers."))


# ----------------------------------------------------------------------------------------
# a11: graph construction / featurisation (model_utils.py:426-593, :606-617)
# ----------------------------------------------------------------------------------------
# atom order of inference/run.py:15-19
ATOM = {a: i for i, a in enumerate(["N", "CA", "C", "O", "OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "O2'", "C1'"])}


def _virtual_atom(p0, p1, p2, wa, wb, wc):
    """get_Cb (model_utils.py:521-526): ideal Cb / N_na from three backbone atoms."""
    b = p1 - p0
    c = p2 - p1
    a = torch.cross(b, c, dim=-1)
    return wa * a + wb * b + wc * c + p1


def knn(P, mask, top_k):
    """_dist (model_utils.py:489-497): masked pairwise distances + top-k smallest."""
    m2 = torch.unsqueeze(mask, 1) * torch.unsqueeze(mask, 2)
    dX = torch.unsqueeze(P, 1) - torch.unsqueeze(P, 2)
    D = m2 * torch.sqrt(torch.sum(dX ** 2, 3) + 1e-6)
    D_max, _ = torch.max(D, -1, keepdim=True)
    D_adj = D + (1. - m2) * D_max
    return torch.topk(D_adj, int(np.minimum(top_k, P.shape[1])), dim=-1, largest=False)


def rbf_all_pairs(X18, E_idx, M18, num_rbf=16):
    """_get_all_rbf + _rbf (model_utils.py:499-519): [B,L,K,18*18*16]."""
    b, l = X18.shape[:2]
    Xg = gather_nodes(X18.reshape(b, l, -1), E_idx)
    Xg = Xg.reshape(list(Xg.shape[:-1]) + list(X18.shape[-2:]))
    D = torch.sqrt(torch.sum((X18[:, :, None, :, None, :] - Xg[:, :, :, None, :, :]) ** 2, -1) + 1e-6)
    mu = torch.linspace(2., 22., num_rbf).view(1, 1, 1, 1, 1, -1)
    sigma = (22. - 2.) / num_rbf
    R = torch.exp(-((torch.unsqueeze(D, -1) - mu) / sigma) ** 2)
    Mg = gather_nodes(M18, E_idx)
    R = R * M18[:, :, None, :, None, None] * Mg[:, :, :, None, :, None]
    return R.view(b, l, E_idx.shape[2], -1)


def positional(w, offset, same_chain, max_rel=32):
    """PositionalEncodings.forward (model_utils.py:613-617)."""
    d = torch.clip(offset + max_rel, 0, 2 * max_rel) * same_chain + (1 - same_chain) * (2 * max_rel + 1)
    return _lin(w, "features.embeddings.linear", F.one_hot(d, 2 * max_rel + 2).float())


def features(w, fd, top_k, na_ref_atom="C1'"):
    """ProteinFeaturesNA.forward in eval mode (model_utils.py:528-593).  na_ref_atom: the nucleic reference atom of the kNN
    point CA + X[na_ref_atom] (ctor argument, model_utils.py:438,457,554,573; na_model_utils.py:361,381,478,497)."""
    X, mask = fd["X"], fd["mask"]
    Ca, N, C = X[:, :, ATOM["CA"], :], X[:, :, ATOM["N"], :], X[:, :, ATOM["C"], :]
    Cb = _virtual_atom(N, Ca, C, -0.58273431, 0.56802827, -0.54067466)
    ref_na = X[:, :, ATOM[na_ref_atom], :]
    N_na = _virtual_atom(X[:, :, ATOM["O4'"], :], X[:, :, ATOM["C1'"], :], X[:, :, ATOM["C2'"], :],
                         -0.56967352, 0.51055973, -0.53122153)
    X18 = torch.cat((X, Cb[:, :, None, :], N_na[:, :, None, :]), -2)
    M18 = torch.cat((fd["X_m"], fd["protein_mask"][:, :, None],
                     (fd["rna_mask"] + fd["dna_mask"])[:, :, None]), -1)
    if w["features.edge_embedding.weight"].shape[1] == 16 + 16 * 17 * 17:
        # include_pred_na_N = 0 (model_utils.py:480-483,555-567; na_model_utils.py:404-407,479-491): no virtual N_na atom,
        # 17 x 17 atom pairs, edge_embedding is [128 x 4640]
        X18, M18 = X18[:, :, :17], M18[:, :, :17]
    _, E_idx = knn(Ca + ref_na, mask, top_k)
    R = rbf_all_pairs(X18, E_idx, M18)
    R_idx, chain = fd["R_idx"], fd["chain_labels"]
    offset = gather_edges((R_idx[:, :, None] - R_idx[:, None, :])[:, :, :, None], E_idx)[:, :, :, 0]
    same = gather_edges(((chain[:, :, None] - chain[:, None, :]) == 0).long()[:, :, :, None], E_idx)[:, :, :, 0]
    E = torch.cat((positional(w, offset.long(), same), R), -1)
    E = _ln(w, "features.norm_edges", _lin(w, "features.edge_embedding", E))
    V = F.one_hot(fd["R_polymer_type"], num_classes=w["features.node_embedding.weight"].shape[1]).float()
    V = _ln(w, "features.norm_nodes", _lin(w, "features.node_embedding", V))
    return V, E, E_idx


# ----------------------------------------------------------------------------------------
# a7: encoder (model_utils.py:71-99)
# ----------------------------------------------------------------------------------------
def encode_from_graph(w, V, E, E_idx, mask):
    h_V = _lin(w, "W_v", V)
    h_E = _lin(w, "W_e", E)
    mask_attend = gather_nodes(mask.unsqueeze(-1), E_idx).squeeze(-1)
    mask_attend = mask.unsqueeze(-1) * mask_attend
    for i in range(n_layers(w, "encoder")):
        h_V, h_E = enc_layer(w, f"encoder_layers.{i}.", h_V, h_E, E_idx, mask, mask_attend)
    return h_V, h_E


def encode(w, fd, top_k, na_ref_atom="C1'"):
    V, E, E_idx = features(w, fd, top_k, na_ref_atom)
    h_V, h_E = encode_from_graph(w, V, E, E_idx, fd["mask"])
    return h_V, h_E, E_idx


# ----------------------------------------------------------------------------------------
# a8/a10: parallel (teacher-forced) decoder
# ----------------------------------------------------------------------------------------
def decoding_order_of(chain_mask, randn):
    """argsort((chain_mask+1e-4)*|randn|)   (model_utils.py:389, na_model_utils.py:623)."""
    return torch.argsort((chain_mask + 0.0001) * torch.abs(randn))


def backward_mask(decoding_order, E_idx):
    """mask_attend[b,i,k] = 1 iff neighbour E_idx[b,i,k] is decoded before i.

    The reference builds this with a one-hot permutation einsum against a strict
    lower-triangular matrix and a gather (model_utils.py:391-393); the values
    are exactly {0.,1.}, so the O(N*K) rank comparison below is bit-identical
    (checked against the einsum in tests/test_oracle_golden.py)."""
    b, n = decoding_order.shape
    rank = torch.empty_like(decoding_order)
    rank.scatter_(1, decoding_order, torch.arange(n).unsqueeze(0).expand(b, -1))
    rb = rank[: E_idx.shape[0]]
    r_j = torch.gather(rb, 1, E_idx.reshape(E_idx.shape[0], -1)).view(E_idx.shape)
    return (r_j < rb.unsqueeze(-1)).float().unsqueeze(-1)


def backward_mask_einsum(decoding_order, E_idx):
    """The reference's literal construction, kept for the equivalence test."""
    L = decoding_order.shape[-1]
    P = F.one_hot(decoding_order, num_classes=L).float()
    om = torch.einsum('ij, biq, bjp->bqp', (1 - torch.triu(torch.ones(L, L))), P, P)
    return torch.gather(om, 2, E_idx).unsqueeze(-1)


def decode_parallel(w, h_V, h_E, E_idx, S, mask, mask_attend_bw):
    """Decoder stack + logits (model_utils.py:406-421 == na_model_utils.py:610-642)."""
    mask_1D = mask.view([mask.size(0), mask.size(1), 1, 1])
    mask_bw = mask_1D * mask_attend_bw
    mask_fw = mask_1D * (1. - mask_attend_bw)
    h_S = F.embedding(S, w["W_s.weight"])
    h_ES = cat_neighbors_nodes(h_S, h_E, E_idx)
    h_EX_enc = cat_neighbors_nodes(torch.zeros_like(h_S), h_E, E_idx)
    h_EXV_enc_fw = mask_fw * cat_neighbors_nodes(h_V, h_EX_enc, E_idx)
    for i in range(n_layers(w, "decoder")):
        h_ESV = cat_neighbors_nodes(h_V, h_ES, E_idx)
        h_ESV = mask_bw * h_ESV + h_EXV_enc_fw
        h_V = dec_layer(w, f"decoder_layers.{i}.", h_V, h_ESV, mask)
    logits = _lin(w, "W_out", h_V)
    return F.log_softmax(logits, dim=-1), logits


def score_from_encoded(w, h_V, h_E, E_idx, S, mask, chain_mask, randn, batch_size=1):
    """ProteinMPNN.score after encode() (model_utils.py:388-424)."""
    chain_mask = mask * chain_mask
    order = decoding_order_of(chain_mask, randn)
    m_att = backward_mask(order, E_idx)          # only row(s) matching E_idx's batch are used
    rep = lambda t: t.repeat(batch_size, *([1] * (t.dim() - 1)))
    log_probs, _ = decode_parallel(w, rep(h_V), rep(h_E), rep(E_idx), rep(S).long(), rep(mask), m_att)
    return {"S": rep(S), "log_probs": log_probs, "decoding_order": order[0]}


def score(w, fd, top_k):
    h_V, h_E, E_idx = encode(w, fd, top_k)
    return score_from_encoded(w, h_V, h_E, E_idx, fd["S"], fd["mask"], fd["chain_mask"], fd["randn"],
                              fd.get("batch_size", 1))


def unconditional_probs(w, fd, top_k):
    """model_utils.py:329-364: decoder sees encoder context only."""
    h_V, h_E, E_idx = encode(w, fd, top_k)
    bs = fd.get("batch_size", 1)
    rep = lambda t: t.repeat(bs, *([1] * (t.dim() - 1)))
    mask = fd["mask"]
    zeros_att = torch.zeros(list(E_idx.shape) + [1])
    mask_fw = mask.view([mask.size(0), mask.size(1), 1, 1]) * (1. - zeros_att)
    h_V, h_E, E_idx, mask_fw, mask = rep(h_V), rep(h_E), rep(E_idx), rep(mask_fw), rep(mask)
    h_EX = cat_neighbors_nodes(torch.zeros_like(h_V), h_E, E_idx)
    ctx = mask_fw * cat_neighbors_nodes(h_V, h_EX, E_idx)
    for i in range(n_layers(w, "decoder")):
        h_V = dec_layer(w, f"decoder_layers.{i}.", h_V, ctx, mask)
    return {"log_probs": F.log_softmax(_lin(w, "W_out", h_V), dim=-1)}


def forward_train(w, fd, top_k, randn, decode_protein_first=False, na_ref_atom="C1'"):
    """Training-copy ProteinMPNN.forward in eval mode with the decoding-order
    noise passed in (na_model_utils.py:589-646; its internal torch.randn is :623)."""
    h_V, h_E, E_idx = encode(w, fd, top_k, na_ref_atom)
    chain_M = fd["mask"]
    if decode_protein_first:
        chain_M = chain_M.masked_fill(fd["protein_mask"].to(torch.bool), 0.0)
    order = decoding_order_of(chain_M, randn)
    log_probs, logits = decode_parallel(w, h_V, h_E, E_idx, fd["S"].long(), fd["mask"],
                                        backward_mask(order, E_idx))
    return log_probs, F.softmax(logits, dim=-1)


# ----------------------------------------------------------------------------------------
# a12: training tail — label-smoothed loss, gradients by torch autograd through the restatement above,
# Noam learning rate (na_model_utils.py:100-146, 648-686; na_run.py:131-154, 198-238)
# ----------------------------------------------------------------------------------------
NO_LOSS_TOKENS = ("UNK", "DX", "RX", "MAS", "PAD")            # na_run.py:131-136
POLYMER_RESTYPES = {"protein": 21, "dna": 5, "rna": 5}        # list lengths of na_data_utils.py:185-223


def restype_masks(restype_to_int, num_letters=33):
    """na_run.py:138-154: 0/1 vocabulary vectors per polymer (+ the reference's divisors 21 / 5 / 5)."""
    prot = [k for k, v in restype_to_int.items() if v <= 20][:21]
    names = {"protein": prot, "dna": ["DA", "DC", "DG", "DT", "DX"], "rna": ["A", "C", "G", "U", "RX"]}
    out = {}
    for key, lst in names.items():
        v = torch.zeros(num_letters)
        v[[restype_to_int[n] for n in lst]] = 1
        out[key] = v
    return out, dict(POLYMER_RESTYPES)


def loss_smoothed(S, log_probs, mask, polymer_masks, restype_mask, restype_num, weight=0.1, tokens=2000.0, num_letters=33,
                  ppm_mask=None, aligned_ppm=None):
    """na_model_utils.py:111-146: fp64 one-hot — replaced by the aligned position-probability row where ppm_mask is set
    (:134, the specificity model's target) —, (1-weight) on the polymer letters, plus weight/num on the letters of the
    residue's own polymer; sum(loss*mask)/tokens."""
    onehot = F.one_hot(S, num_letters).to(torch.float64)
    if ppm_mask is not None:
        onehot[ppm_mask.bool()] = aligned_ppm[ppm_mask.bool()]
    eps = sum(polymer_masks[k][:, :, None] * restype_mask[k][None, None, :] * (weight / restype_num[k])
              for k in ("protein", "dna", "rna"))
    allm = restype_mask["protein"] + restype_mask["dna"] + restype_mask["rna"]
    onehot[:, :, allm.bool()] *= (1 - weight)
    onehot = onehot + eps
    loss = -(onehot * log_probs).sum(-1)
    return loss, torch.sum(loss * mask) / tokens


def noam_rate(step, model_size=128, factor=2, warmup=4000):
    """NoamOpt.rate (na_model_utils.py:672-678)."""
    return factor * (model_size ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


def train_loss_and_grads(w, fd, top_k, randn, restype_to_int, weight=0.1, tokens=2000.0, ppm_mask=None, aligned_ppm=None):
    """One training forward/backward of na_run.py:198-238 (dropout 0, no coordinate noise) on the restatement:
    returns loss (fp64 scalar), log_probs and {key: grad}."""
    wg = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    with torch.enable_grad():
        log_probs, _ = forward_train(wg, fd, top_k, randn)
        S = fd["S"].long()
        no_loss = torch.tensor([restype_to_int[t] for t in NO_LOSS_TOKENS])
        S_mask = 1 - torch.any(S[:, :, None] == no_loss[None, None, :], dim=-1).long()
        rm, rn = restype_masks(restype_to_int, log_probs.shape[-1])
        pm = {"protein": fd["protein_mask"], "dna": fd["dna_mask"], "rna": fd["rna_mask"]}
        _, loss = loss_smoothed(S, log_probs, fd["mask"] * S_mask, pm, rm, rn, weight, tokens, log_probs.shape[-1],
                                ppm_mask=ppm_mask, aligned_ppm=aligned_ppm)
        loss.backward()
    return loss.detach(), log_probs.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in wg.items()}


# ----------------------------------------------------------------------------------------
# a9: autoregressive sampler, non-symmetric branch (model_utils.py:126-218)
# ----------------------------------------------------------------------------------------
SPECIAL_TOKENS = (20, 25, 30, 31, 32)   # UNK, DX, RX, MAS, PAD (run.py:32-66)


def sample(w, fd, top_k, special=SPECIAL_TOKENS, S_forced=None):
    """Step-by-step decoder.  Consumes torch's global CPU RNG exactly like the
    reference (one ``torch.multinomial`` per step) so that, under the same
    ``torch.manual_seed``, it reproduces the reference draw-for-draw.  With
    ``S_forced`` given, the draw is replaced by that sequence (teacher forcing)."""
    bs = fd["batch_size"]
    S_true, mask, bias, T = fd["S"], fd["mask"], fd["bias"], fd["temperature"]
    pair_bias = fd.get("pair_bias")
    B, L = S_true.shape
    nl = w["W_out.weight"].shape[0]
    h_V, h_E, E_idx = encode(w, fd, top_k)
    chain_mask = mask * fd["chain_mask"]
    order = decoding_order_of(chain_mask, fd["randn"])
    E_idx = E_idx.repeat(bs, 1, 1)
    m_att = backward_mask(order, E_idx)
    m1 = mask.view([B, L, 1, 1])
    mask_bw, mask_fw = m1 * m_att, m1 * (1. - m_att)
    S_true, h_V, h_E = S_true.repeat(bs, 1), h_V.repeat(bs, 1, 1), h_E.repeat(bs, 1, 1, 1)
    chain_mask, mask, bias = chain_mask.repeat(bs, 1), mask.repeat(bs, 1), bias.repeat(bs, 1, 1)
    if pair_bias is not None:
        pair_bias = pair_bias.repeat(bs, 1, 1, 1, 1)
    all_probs = torch.zeros((bs, L, nl))
    all_logp = torch.zeros((bs, L, nl))
    h_S = torch.zeros_like(h_V)
    S = (nl - 1) * torch.ones((bs, L), dtype=torch.int64)
    nd = n_layers(w, "decoder")
    stack = [h_V] + [torch.zeros_like(h_V) for _ in range(nd)]
    h_EX = cat_neighbors_nodes(torch.zeros_like(h_S), h_E, E_idx)
    ctx_fw = mask_fw * cat_neighbors_nodes(h_V, h_EX, E_idx)
    for t_ in range(L):
        t = order[:, t_]
        g1 = lambda a: torch.gather(a, 1, t[:, None])[:, 0]
        g4 = lambda a: torch.gather(a, 1, t[:, None, None, None].repeat(1, 1, a.shape[-2], a.shape[-1]))
        g3 = lambda a: torch.gather(a, 1, t[:, None, None].repeat(1, 1, a.shape[-1]))
        cm_t, mask_t = g1(chain_mask), g1(mask)
        bias_t = g3(bias)[:, 0, :]
        E_idx_t = g3(E_idx)
        h_ES_t = cat_neighbors_nodes(h_S, g4(h_E), E_idx_t)
        ctx_t, bw_t = g4(ctx_fw), g4(mask_bw)
        for l in range(nd):
            h_ESV_t = bw_t * cat_neighbors_nodes(stack[l], h_ES_t, E_idx_t) + ctx_t
            out = dec_layer(w, f"decoder_layers.{l}.", g3(stack[l]), h_ESV_t, mask_V=mask_t)
            stack[l + 1].scatter_(1, t[:, None, None].repeat(1, 1, H), out)
        logits = _lin(w, "W_out", g3(stack[-1])[:, 0])
        logp = F.log_softmax(logits, dim=-1)
        if pair_bias is not None:          # model_utils.py:169-172: sum_j pair_bias[t, :, j, S_j] with the running S
            pb = torch.gather(pair_bias, 1, t[:, None, None, None, None].repeat(1, 1, nl, L, nl))[:, 0]
            pb = torch.gather(pb, -1, S[:, None, :, None].repeat(1, nl, 1, 1))[:, :, :, 0].sum(-1)
            probs = F.softmax((logits + bias_t + pb) / T, dim=-1)
        else:
            probs = F.softmax((logits + bias_t) / T, dim=-1)
        for tok in special:
            probs[:, tok] = 0
        probs = probs / torch.sum(probs, dim=-1, keepdim=True)
        if S_forced is None:
            S_t = torch.multinomial(probs, 1)[:, 0]
        else:
            S_t = g1(S_forced)
        # the reference scatters probabilities with an index of nl-1 columns (model_utils.py:211)
        all_probs.scatter_(1, t[:, None, None].repeat(1, 1, nl - 1), (cm_t[:, None, None] * probs[:, None, :]).float())
        all_logp.scatter_(1, t[:, None, None].repeat(1, 1, nl), (cm_t[:, None, None] * logp[:, None, :]).float())
        S_t = (S_t * cm_t + g1(S_true) * (1.0 - cm_t)).long()
        h_S.scatter_(1, t[:, None, None].repeat(1, 1, H), F.embedding(S_t, w["W_s.weight"])[:, None, :])
        S.scatter_(1, t[:, None], S_t[:, None])
    return {"S": S, "sampling_probs": all_probs, "log_probs": all_logp, "decoding_order": order}


def sample_symmetric(w, fd, top_k, special=SPECIAL_TOKENS, S_forced=None):
    """Symmetry-tied branch of ProteinMPNN.sample (model_utils.py:219-326): tied residues are visited together,
    their logits are summed with the given weights and ONE token is drawn per group (same RNG consumption as the
    reference: one torch.multinomial per group)."""
    bs = fd["batch_size"]
    S_true, mask, bias, T = fd["S"], fd["mask"], fd["bias"], fd["temperature"]
    groups_in, weights_in = fd["symmetry_residues"], fd["symmetry_weights"]
    B, L = S_true.shape
    nl = w["W_out.weight"].shape[0]
    h_V, h_E, E_idx = encode(w, fd, top_k)
    chain_mask = mask * fd["chain_mask"]
    order = decoding_order_of(chain_mask, fd["randn"])
    sym_w = torch.ones([L], dtype=torch.float32)
    for i1, grp in enumerate(groups_in):
        for i2, item in enumerate(grp):
            sym_w[item] = weights_in[i1][i2]
    new_order = []
    for t_dec in list(order[0].numpy()):
        if t_dec not in list(itertools.chain(*new_order)):
            hit = [g for g in groups_in if t_dec in g]
            new_order.append(hit[0] if hit else [t_dec])
    order = torch.tensor(list(itertools.chain(*new_order)))[None].repeat(B, 1)
    m_att = backward_mask(order, E_idx)
    m1 = mask.view([B, L, 1, 1])
    mask_bw, mask_fw = m1 * m_att, m1 * (1. - m_att)
    rep = lambda a: a.repeat(bs, *([1] * (a.dim() - 1)))
    S_true, h_V, h_E, E_idx = rep(S_true), rep(h_V), rep(h_E), rep(E_idx)
    mask_fw, mask_bw, chain_mask, mask, bias = rep(mask_fw), rep(mask_bw), rep(chain_mask), rep(mask), rep(bias)
    pair_bias = fd.get("pair_bias")
    if pair_bias is not None:
        pair_bias = rep(pair_bias)
    all_probs = torch.zeros((bs, L, nl))
    all_logp = torch.zeros((bs, L, nl))
    h_S = torch.zeros_like(h_V)
    S = (nl - 1) * torch.ones((bs, L), dtype=torch.int64)
    nd = n_layers(w, "decoder")
    stack = [h_V] + [torch.zeros_like(h_V) for _ in range(nd)]
    h_EX = cat_neighbors_nodes(torch.zeros_like(h_S), h_E, E_idx)
    ctx_fw = mask_fw * cat_neighbors_nodes(h_V, h_EX, E_idx)
    for t_list in new_order:
        total = 0.0
        for t in t_list:
            cm_t, mask_t, bias_t = chain_mask[:, t], mask[:, t], bias[:, t]
            E_idx_t, h_E_t = E_idx[:, t:t + 1], h_E[:, t:t + 1]
            h_ES_t = cat_neighbors_nodes(h_S, h_E_t, E_idx_t)
            for l in range(nd):
                h_ESV_t = mask_bw[:, t:t + 1] * cat_neighbors_nodes(stack[l], h_ES_t, E_idx_t) + ctx_fw[:, t:t + 1]
                stack[l + 1][:, t:t + 1, :] = dec_layer(w, f"decoder_layers.{l}.", stack[l][:, t:t + 1], h_ESV_t,
                                                        mask_V=mask_t[:, None])
            logits = _lin(w, "W_out", stack[-1][:, t])
            all_logp[:, t] = (cm_t[:, None] * F.log_softmax(logits, dim=-1)).float()
            total = total + sym_w[t] * logits
            if pair_bias is not None:      # model_utils.py:273-276: the row of member t with the running S; the LAST member's is used (:300)
                pb = torch.gather(pair_bias[:, t], -1, S[:, None, :, None].repeat(1, nl, 1, 1))[:, :, :, 0].sum(-1)
        probs = F.softmax((total + bias_t + (pb if pair_bias is not None else 0.0)) / T, dim=-1)
        for tok in special:
            probs[:, tok] = 0
        probs = probs / torch.sum(probs, dim=-1, keepdim=True)
        S_t = torch.multinomial(probs, 1)[:, 0] if S_forced is None else None
        for t in t_list:
            cm_t = chain_mask[:, t]
            all_probs[:, t] = (cm_t[:, None] * probs).float()
            if S_forced is not None:
                S_t = S_forced[:, t]
            S_t = (S_t * cm_t + S_true[:, t] * (1.0 - cm_t)).long()
            h_S[:, t] = F.embedding(S_t, w["W_s.weight"])
            S[:, t] = S_t
    return {"S": S, "sampling_probs": all_probs, "log_probs": all_logp, "decoding_order": order.repeat(bs, 1)}


# ----------------------------------------------------------------------------------------
# helpers used by the bench's cpu_baseline leg and the tests
# ----------------------------------------------------------------------------------------
def encdec_from_graph(w, V, E, E_idx, S, mask, chain_mask, randn):
    """The BASELINE metric scope: (V,E,E_idx,...) -> log_probs."""
    h_V, h_E = encode_from_graph(w, V, E, E_idx, mask)
    return score_from_encoded(w, h_V, h_E, E_idx, S, mask, chain_mask, randn)


def to_torch(d):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
            for k, v in d.items()}
