"""The CPU oracle (oracle/cpu_ref.py) against the committed golden vectors.

The goldens were produced by the real reference in the build container
(oracle/make_goldens.py).  Bit-exact equality is required: the oracle performs
the same ATen op sequence as the reference on CPU.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from na_mpnn_amd import synth
from oracle import cpu_ref

torch.set_grad_enabled(False)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)


def tw(weights):
    return {k: torch.from_numpy(v) for k, v in weights.items()}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def exact(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == b.shape
    assert np.array_equal(a, b), f"max|d|={np.abs(a.astype(np.float64) - b).max()}"


def test_g1_gather(golden_dir):
    g = load(golden_dir, "g1_gather")
    t = torch.from_numpy
    idx = t(g["idx"].astype(np.int64))
    exact(cpu_ref.gather_nodes(t(g["nodes"]), idx), g["gather_nodes"])
    exact(cpu_ref.gather_edges(t(g["edges"]), idx), g["gather_edges"])
    exact(cpu_ref.cat_neighbors_nodes(t(g["nodes"]), t(g["nbrs"]), idx), g["cat"])


def test_g2_layers(golden_dir, weights_np):
    g = load(golden_dir, "g2_layers")
    x = synth.make_graph(seed=202, batch=1, n=128, k=48, masked_frac=0.1)
    assert np.array_equal(digest(x["V"], x["E"], x["E_idx"], x["mask"]), g["in_digest"]), \
        "synthetic input generator drifted from the one the goldens were made with"
    t = {k: torch.from_numpy(v) for k, v in x.items()}
    w = tw(weights_np)
    E_idx, mask = t["E_idx"].long(), t["mask"]
    m_att = cpu_ref.gather_nodes(mask.unsqueeze(-1), E_idx).squeeze(-1) * mask.unsqueeze(-1)
    hV, hE = cpu_ref.enc_layer(w, "encoder_layers.1.", t["V"], t["E"], E_idx, mask, m_att)
    exact(hV[0], g["enc_hV"])
    exact(hE[0, ::16], g["enc_hE_rows"])
    ctx = np.random.default_rng(203).standard_normal((1, 128, 48, 384)).astype(np.float32)
    exact(cpu_ref.dec_layer(w, "decoder_layers.2.", t["V"], torch.from_numpy(ctx), mask)[0], g["dec_hV"])


@pytest.mark.parametrize("n,tag,mf,batch", [(256, "n256", 0.05, 1), (40, "n40_LltK", 0.0, 1),
                                             (200, "b3_n200", 0.1, 3), (1000, "n1000", 0.0, 1)])
def test_g3_encdec(golden_dir, weights_np, n, tag, mf, batch):
    g = load(golden_dir, f"g3_encdec_{tag}")
    x = synth.make_graph(seed=300 + n + batch, batch=batch, n=n, k=48, masked_frac=mf)
    assert np.array_equal(digest(x["V"], x["E"], x["E_idx"], x["S"], x["mask"], x["randn"]), g["in_digest"])
    t = {k: torch.from_numpy(v) for k, v in x.items()}
    w = tw(weights_np)
    E_idx = t["E_idx"].long()
    h_V, h_E = cpu_ref.encode_from_graph(w, t["V"], t["E"], E_idx, t["mask"])
    stride = int(g["row_stride"])
    exact(h_V[:, ::stride], g["enc_hV_layers"][-1])
    exact(h_E[:, ::max(1, n // 16)][:, :16], g["enc_hE_rows"])
    for b in range(batch):
        o = cpu_ref.score_from_encoded(w, h_V[b:b + 1], h_E[b:b + 1], E_idx[b:b + 1], t["S"][b:b + 1],
                                       t["mask"][b:b + 1], t["chain_mask"][b:b + 1], t["randn"][b:b + 1])
        exact(o["log_probs"][0], g["log_probs"][b])
        assert np.array_equal(o["decoding_order"].numpy(), g["decoding_order"][b])
        assert np.array_equal(o["log_probs"][0].argmax(-1).numpy(), g["argmax"][b])


@pytest.mark.parametrize("n,k,tag,kw", [(97, 32, "n97_k32", dict(missing_atom_frac=0.05, masked_frac=0.04)),
                                         (150, 48, "n150_k48", {}), (32, 48, "n32_k48_LltK", {})])
def test_g4_from_coordinates(golden_dir, weights_np, n, k, tag, kw):
    g = load(golden_dir, f"g4_fromX_{tag}")
    cx = synth.make_complex(seed=400 + n, n=n, **kw)
    assert np.array_equal(digest(*[cx[k_] for k_ in sorted(cx)]), g["in_digest"])
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None] for k_, v in cx.items()}
    fd["batch_size"] = 1
    w = tw(weights_np)
    V, E, E_idx = cpu_ref.features(w, fd, k)
    assert np.array_equal(E_idx[0].numpy(), g["E_idx"].astype(np.int64))
    exact(V[0], g["V"])
    exact(E[0, ::max(1, n // 8)][:8], g["E_rows"])
    sc = cpu_ref.score(w, fd, k)
    exact(sc["log_probs"][0], g["log_probs"])
    assert np.array_equal(sc["decoding_order"].numpy(), g["decoding_order"])
    exact(cpu_ref.unconditional_probs(w, fd, k)["log_probs"][0], g["uncond_log_probs"])
    fdt = dict(fd); fdt["S"] = fd["S"].long()
    lp, _ = cpu_ref.forward_train(w, fdt, k, torch.from_numpy(g["train_randn"]))
    exact(lp[0], g["train_log_probs"])


def test_g5_sample_teacher_forced(golden_dir, weights_np):
    """model_utils.py:367: score(S_sampled).log_probs == sample().log_probs on designed positions."""
    g = load(golden_dir, "g5_sample")
    n, k, bs = 60, 16, 3
    cx = synth.make_complex(seed=500, n=n, n_chains=2)
    cx["chain_mask"][:7] = 0
    assert np.array_equal(digest(*[cx[k_] for k_ in sorted(cx)]), g["in_digest"])
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None] for k_, v in cx.items()}
    fd.update({"batch_size": bs, "temperature": 0.5, "bias": torch.zeros(1, n, 33),
               "symmetry_residues": [[]], "symmetry_weights": [[]], "randn": torch.from_numpy(g["randn"])})
    w = tw(weights_np)
    S = torch.from_numpy(g["S"].astype(np.int64))
    o = cpu_ref.sample(w, fd, k, S_forced=S)
    exact(o["log_probs"], g["log_probs"])
    exact(o["sampling_probs"], g["sampling_probs"])
    assert np.array_equal(o["decoding_order"].numpy(), g["decoding_order"])
    # the reference's own stated invariant, per sample row
    cm = (fd["mask"] * fd["chain_mask"])[0].bool()
    for b in range(bs):
        fdb = dict(fd); fdb["batch_size"] = 1; fdb["S"] = S[b:b + 1]; fdb["randn"] = fd["randn"][b:b + 1]
        sc = cpu_ref.score(w, fdb, k)
        d = (sc["log_probs"][0][cm] - torch.from_numpy(g["log_probs"][b])[cm]).abs().max()
        assert d < 2e-5, float(d)


def test_rank_compare_equals_reference_einsum():
    rng = np.random.default_rng(7)
    L, K = 37, 9
    order = torch.from_numpy(np.stack([rng.permutation(L) for _ in range(2)]))
    E_idx = torch.from_numpy(rng.integers(0, L, (2, L, K)))
    assert torch.equal(cpu_ref.backward_mask(order, E_idx), cpu_ref.backward_mask_einsum(order, E_idx))


G6_GROUPS = [[0, 5, 9], [12, 13], [21, 30, 31, 32]]
G6_WEIGHTS = [[1.0, 0.5, 2.0], [1.0, -1.0], [0.7, 0.7, 0.7, 0.7]]


def g6_inputs(g):
    from na_mpnn_amd.cli import make_pair_bias
    n, k, bs = 40, 16, 2
    cx = synth.make_complex(seed=600, n=n, n_chains=2)
    cx["chain_mask"][[3, 21]] = 0
    assert np.array_equal(digest(*[cx[k_] for k_ in sorted(cx)]), g["in_digest"])
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None] for k_, v in cx.items()}
    fd.update({"batch_size": bs, "temperature": 0.7, "bias": torch.zeros(1, n, 33), "randn": torch.from_numpy(g["randn"])})
    fds = dict(fd); fds.update({"symmetry_residues": G6_GROUPS, "symmetry_weights": G6_WEIGHTS})
    fdp = dict(fd); fdp.update({"symmetry_residues": [[]], "symmetry_weights": [[]],
                                "pair_bias": make_pair_bias(fd["chain_labels"][0], fd["R_idx"][0], torch.from_numpy(g["pair_bias_AA"]))})
    return cx, k, fds, fdp


def test_g6_symmetric_and_pair_bias_sampling(golden_dir, weights_np):
    """model_utils.py:219-326 (symmetry-tied groups) and :169-172 (pair_bias), teacher-forced with the reference's draws."""
    g = load(golden_dir, "g6_sample_variants")
    cx, k, fds, fdp = g6_inputs(g)
    w = tw(weights_np)
    o = cpu_ref.sample_symmetric(w, fds, k, S_forced=torch.from_numpy(g["sym_S"].astype(np.int64)))
    assert np.array_equal(o["decoding_order"].numpy(), g["sym_order"])
    exact(o["log_probs"], g["sym_log_probs"]); exact(o["sampling_probs"], g["sym_probs"])
    assert np.array_equal(o["S"].numpy(), g["sym_S"])
    for grp in G6_GROUPS:      # one draw per group; a fixed member overrides the running token for the members after it
        for t_prev, t in zip(grp[:-1], grp[1:]):
            if cx["chain_mask"][t] and cx["mask"][t]:
                assert (g["sym_S"][:, t] == g["sym_S"][:, t_prev]).all()
    p = cpu_ref.sample(w, fdp, k, S_forced=torch.from_numpy(g["pb_S"].astype(np.int64)))
    exact(p["log_probs"], g["pb_log_probs"]); exact(p["sampling_probs"], g["pb_probs"])


def test_g6b_symmetric_with_pair_bias(golden_dir, weights_np):
    """Symmetry-tied groups together with pair_bias (model_utils.py:273-276, :300-303), teacher-forced with the reference's draws."""
    g = load(golden_dir, "g6b_symmetric_pair_bias")
    cx, k, fds, fdp = g6_inputs(g)
    fds["pair_bias"] = fdp["pair_bias"]
    o = cpu_ref.sample_symmetric(tw(weights_np), fds, k, S_forced=torch.from_numpy(g["S"].astype(np.int64)))
    assert np.array_equal(o["decoding_order"].numpy(), g["order"]) and np.array_equal(o["S"].numpy(), g["S"])
    exact(o["log_probs"], g["log_probs"]); exact(o["sampling_probs"], g["probs"])


def g7_inputs():
    n, n2, k = 72, 55, 24
    a = synth.make_complex(seed=700, n=n, n_chains=3, masked_frac=0.04)
    b = synth.make_complex(seed=701, n=n2, n_chains=2)
    fd = {}
    for key in a:
        pad = np.zeros((n - n2,) + b[key].shape[1:], b[key].dtype)
        fd[key] = torch.from_numpy(np.stack([a[key], np.concatenate([b[key], pad])]))
    fd["S"] = fd["S"].long()
    fd["S"][1, n2:] = 32
    return fd, k


def g7b_ppm(fd, seed=702, frac=0.4):
    """The PPM targets of golden G7b (same recipe as oracle/make_goldens.py:g7b_ppm)."""
    rng = np.random.default_rng(seed)
    dna, rna, mask = fd["dna_mask"].numpy(), fd["rna_mask"].numpy(), fd["mask"].numpy()
    B, L = mask.shape
    ppm_mask = (((dna + rna) * mask) * (rng.random((B, L)) < frac)).astype(np.int64)
    ppm = np.zeros((B, L, 33), np.float64)
    w4 = rng.dirichlet(np.ones(4), size=(B, L))
    ppm[:, :, 21:25] = w4 * dna[:, :, None]
    ppm[:, :, 26:30] = w4 * rna[:, :, None]
    return torch.from_numpy(ppm_mask), torch.from_numpy(ppm)


@pytest.mark.parametrize("tag", ["g7_training", "g7b_training"])
def test_g7_training_step(golden_dir, weights_np, tag):
    """a12: loss and gradients of one training step (na_run.py:198-238) — the reference ran in train mode through
    torch.utils.checkpoint; the oracle's plain autograd must give the same numbers.  g7b: with a non-empty ppm_mask
    (loss_smoothed's position-probability target, na_model_utils.py:134)."""
    from na_mpnn_amd import spec
    g = load(golden_dir, tag)
    fd, k = g7_inputs()
    pm, ppm = g7b_ppm(fd) if tag == "g7b_training" else (None, None)
    loss, lp, grads = cpu_ref.train_loss_and_grads(tw(weights_np), fd, k, torch.from_numpy(g["randn"]), spec.restype_to_int(),
                                                   ppm_mask=pm, aligned_ppm=ppm)
    exact(lp, g["log_probs"])
    assert float(loss) == float(g["loss"])
    names = [str(n) for n in g["names"]]
    assert sorted(grads) == names
    for i, n in enumerate(names):
        got = grads[n].reshape(-1)[torch.from_numpy(g["pick"][i])].numpy()
        assert np.array_equal(got, g["grad_pick"][i]), n
        assert abs(float(grads[n].double().norm()) - g["grad_norm"][i]) <= 1e-12 * max(1.0, g["grad_norm"][i])
    assert abs(cpu_ref.noam_rate(1) - float(g["lr_step1"])) < 1e-18



def test_g4b_no_pred_na_N(golden_dir):
    """include_pred_na_N = 0 (na_model_utils.py:404-407,479-491): 17-atom featurisation + the training copy's forward."""
    from na_mpnn_amd import synth
    g = load(golden_dir, "g4b_noN_n60_k24")
    w = tw(synth.make_weights_noN(0))
    assert w["features.edge_embedding.weight"].shape == (128, 16 + 16 * 17 * 17)
    cx = synth.make_complex(seed=460, n=60, n_chains=3, masked_frac=0.04, missing_atom_frac=0.03)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in cx.items()}
    fd["S"] = fd["S"].long()
    V, E, E_idx = cpu_ref.features(w, fd, 24)
    assert np.array_equal(E_idx[0].numpy(), g["E_idx"].astype(np.int64))
    exact(E[0, ::max(1, 60 // 8)][:8], g["E_rows"])
    lp, _ = cpu_ref.forward_train(w, fd, 24, torch.from_numpy(g["randn"]))
    exact(lp[0], g["log_probs"])


def test_g4c_ctor_variants(golden_dir, weights_np):
    """decode_protein_first=1 (na_model_utils.py:620-621) and na_ref_atom="P" (na_model_utils.py:497; model_utils.py:573): the
    oracle against the outputs of the reference's training copy built with those constructor arguments."""
    from na_mpnn_amd import synth
    g = load(golden_dir, "g4c_ctor_variants_n90_k24")
    w = tw(weights_np)
    cx = synth.make_complex(seed=470, n=90, n_chains=4, masked_frac=0.03, missing_atom_frac=0.03)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in cx.items()}
    fd["S"] = fd["S"].long()
    randn = torch.from_numpy(g["randn"])
    lp, _ = cpu_ref.forward_train(w, fd, 24, randn, decode_protein_first=True)
    exact(lp[0], g["dpf_log_probs"])
    chain_M = fd["mask"].masked_fill(fd["protein_mask"].to(torch.bool), 0.0)
    assert np.array_equal(cpu_ref.decoding_order_of(chain_M, randn)[0].numpy(), g["dpf_decoding_order"])
    lp0, _ = cpu_ref.forward_train(w, fd, 24, randn)
    assert float((lp0[0] - torch.from_numpy(g["dpf_log_probs"])).abs().max()) > 1e-3          # the flag really changes the result
    V, E, E_idx = cpu_ref.features(w, fd, 24, na_ref_atom="P")
    assert np.array_equal(E_idx[0].numpy(), g["refP_E_idx"].astype(np.int64))
    exact(E[0, ::max(1, 90 // 8)][:8], g["refP_E_rows"])
    lp, _ = cpu_ref.forward_train(w, fd, 24, randn, na_ref_atom="P")
    exact(lp[0], g["refP_log_probs"])

