"""GPU tests of the drop-in Python surface (na_mpnn_amd.model.ProteinMPNN) against the goldens the
real reference produced from coordinates (G4/G6) and against the oracle."""
import os

import numpy as np
import pytest
import torch

from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
from oracle import cpu_ref
from featurize_torch import featurize_torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def make_model(weights_np, k, dev):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in weights_np.items()})
    return m.to(dev).eval()


def fd_of(cx, dev):
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k, v in cx.items()}
    fd["batch_size"] = 1
    return fd


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def test_split_bf16_and_exact_fp32_modes_agree(weights_np):
    """The default evaluation of the per-edge and featuriser GEMMs (split-bf16 products) against exact fp32 MFMA on the
    same complex: edge features, log-probs and arg-max — the two must differ by far less than the 1e-3 parity bar."""
    dev = torch.device("cuda:0")
    cx = synth.make_complex(seed=77, n=300, missing_atom_frac=0.03)
    fd = fd_of(cx, dev)
    m = make_model(weights_np, 48, dev)
    out = {}
    for prec in ("x3", "fp32"):
        m.message_precision = prec
        out[prec] = (m.featurize(fd)[1].clone(), m.score(fd)["log_probs"].clone())
    assert maxdiff(out["x3"][0], out["fp32"][0].cpu()) < 1e-4            # measured 2.5e-5 at N=1000
    assert maxdiff(out["x3"][1], out["fp32"][1].cpu()) < 2e-4            # measured 2e-5
    assert torch.equal(out["x3"][1].argmax(-1), out["fp32"][1].argmax(-1))


@pytest.mark.parametrize("prec", ["x3", "fp32"])
@pytest.mark.parametrize("n,k,tag,kw", [(97, 32, "n97_k32", dict(missing_atom_frac=0.05, masked_frac=0.04)),
                                         (150, 48, "n150_k48", {}), (32, 48, "n32_k48_LltK", {})])
def test_score_from_coordinates(golden_dir, weights_np, n, k, tag, kw, prec):
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, f"g4_fromX_{tag}.npz"))
    cx = synth.make_complex(seed=400 + n, n=n, **kw)
    fd = fd_of(cx, dev)
    m = make_model(weights_np, k, dev)
    m.message_precision = prec
    V, E, E_idx = m.featurize(fd)
    # neighbour sets of every unmasked residue must agree with the reference.  (A masked residue's row of
    # the distance matrix is all-equal, so torch.topk's pick there is an arbitrary tie-break on either
    # device — model_utils.py:493-496 — and it cannot influence any unmasked output.)
    valid = cx["mask"].astype(bool)
    ref_idx = np.sort(g["E_idx"].astype(np.int64), -1)
    assert np.array_equal(np.sort(E_idx[0].cpu().numpy(), -1)[valid], ref_idx[valid])
    assert maxdiff(V[0], g["V"]) < 1e-5
    # a11: edge features of the HIP featuriser vs the reference rows stored in the golden.  The golden's rows are in the
    # reference's neighbour order; ours are permuted into that order first (neighbour sets are equal, see above), so the
    # comparison never depends on how either device broke a distance tie.  Masked residues are skipped (their
    # neighbour list is arbitrary on both sides).
    rows = np.arange(n)[::max(1, n // 8)][:8]
    ours_idx, ref_rows_idx = E_idx[0].cpu().numpy(), g["E_idx"].astype(np.int64)
    compared = 0
    for q, i in enumerate(rows):
        if not valid[i]:
            continue
        pos = {int(j): c for c, j in enumerate(ours_idx[i])}
        perm = torch.tensor([pos[int(j)] for j in ref_rows_idx[i]], device=dev)
        assert maxdiff(E[0, i][perm], g["E_rows"][q]) < 2e-4
        compared += 1
    assert compared >= 4
    Vt, Et, It = featurize_torch(m, fd)
    assert torch.equal(torch.sort(It[0][valid], -1)[0], torch.sort(E_idx[0][valid], -1)[0])
    if torch.equal(It[0][valid], E_idx[0][valid]):
        assert maxdiff(E[0][valid], Et[0][valid].cpu()) < 2e-4
    out = m.score(fd)
    assert np.array_equal(out["decoding_order"].cpu().numpy(), g["decoding_order"])
    d = maxdiff(out["log_probs"][0], g["log_probs"])
    assert d < 1e-3, d
    assert np.array_equal(out["log_probs"][0].argmax(-1).cpu().numpy()[valid], g["log_probs"].argmax(-1)[valid])
    up = m.unconditional_probs(fd)
    assert maxdiff(up["log_probs"][0], g["uncond_log_probs"]) < 1e-3
    # training-copy surface with the stored decoding noise
    lp, p = m.forward(fd, decoding_randn=torch.from_numpy(g["train_randn"]).to(dev))
    assert maxdiff(lp[0], g["train_log_probs"]) < 1e-3
    assert maxdiff(p.sum(-1), torch.ones(1, n)) < 1e-5
    # positional wrapper == dict surface
    lp2 = m.forward_positional(fd["X"], fd["S"], fd["mask"], fd["chain_mask"], fd["R_idx"], fd["chain_labels"],
                               fd["randn"], X_m=fd["X_m"], protein_mask=fd["protein_mask"], dna_mask=fd["dna_mask"],
                               rna_mask=fd["rna_mask"], R_polymer_type=fd["R_polymer_type"])
    assert torch.equal(lp2, out["log_probs"])


@pytest.mark.parametrize("prec", ["x3", "fp32"])
@pytest.mark.parametrize("n,k,bs,kw", [(97, 32, 1, dict(frac_protein=0.0, frac_dna=0.0, n_chains=1)), (1000, 48, 1, dict(n_chains=4)),
                                       (300, 30, 3, dict(missing_atom_frac=0.05, masked_frac=0.05)), (257, 70, 1, dict(n_chains=5))])
def test_two_part_featuriser_matches_the_single_launch(weights_np, n, k, bs, kw, prec):
    """namp_featurize with its edge-feature launch split in two parts (each walks every second atom-pair chunk; feat_finish_kernel adds the partial
    rows, normalises and embeds — batches of at most one round of the chip; two to four parts) against the single launch: E and h_E within
    2e-5 / 5e-5 (another order of fp32 sums in front of the LayerNorm), neighbour lists identical; with E and h_E both requested (partial rows in the output
    buffers themselves) and with h_E only (part 1 in the workspace's tail); one-residue workgroups (97 residues), K % 16 != 0, a padded batch."""
    from na_mpnn_amd import hip, shard
    dev = torch.device("cuda:0")
    cxs = [synth.make_complex(seed=9100 + n + i, n=n - 37 * i, **kw) for i in range(bs)]
    fd = fd_of(cxs[0], dev) if bs == 1 else dict(shard.pad_batch(cxs, device=dev), batch_size=1)
    m = make_model(weights_np, k, dev)
    m.message_precision = prec
    L = hip.lib()
    prev = L.namp_set_bf16p(11)                   # single launch
    try:
        _, E0, hE0, I0 = m._featurize_hip(fd, want_E=True, want_hE=True)
        E0, hE0 = E0.clone(), hE0.clone()
        diffs = []
        for parts in (2, 3, 4):
            L.namp_set_bf16p(11 | 32 | ((parts - 2) << 6))
            _, E1, hE1, I1 = m._featurize_hip(fd, want_E=True, want_hE=True)
            E1, hE1 = E1.clone(), hE1.clone()
            _, _, hE2, I2 = m._featurize_hip(fd, want_E=False, want_hE=True)
            hE2 = hE2.clone()
            _, E3, _, I3 = m._featurize_hip(fd, want_E=True, want_hE=False)      # E only: part 0 in E, the others in the workspace
            assert torch.equal(I0, I1) and torch.equal(I0, I2) and torch.equal(I0, I3)
            assert torch.isfinite(E1).all() and torch.isfinite(hE1).all() and torch.isfinite(hE2).all() and torch.isfinite(E3).all()
            valid = fd["mask"].bool()
            assert torch.equal(E3[valid], E1[valid])                               # the same sums whatever buffers hold the partial rows
            diffs.append(tuple(float((x - y)[valid].abs().max()) for x, y in ((E0, E1), (hE0, hE1), (hE0, hE2))))
    finally:
        L.namp_set_bf16p(prev)
    for parts, (d_e, d_h, d_h2) in zip((2, 3, 4), diffs):
        print(f"featuriser in {parts} parts: max|dE| = {d_e:.1e}, max|dh_E| = {d_h:.1e} / {d_h2:.1e}")
        # measured 5e-6 / 1.2e-5: fp32 rounding of a 5,200-term sum taken in another order, in front of a LayerNorm; the parity bars are 2e-4 / 1e-3
        assert d_e < 2e-5 and d_h < 5e-5 and d_h2 < 5e-5, (parts, d_e, d_h, d_h2)


def test_score_from_coordinates_at_the_headline_size(weights_np):
    """score() FROM COORDINATES at the size the metric is quoted on (BASELINE configs[1]: N = 1000, K = 48; the complex behind bench.py's
    `gpu_full_forward_from_X`: synth.make_complex(seed=77, n=1000, n_chains=4)) against the CPU oracle's score() on the same feature dict —
    featuriser, encoder and decoder in one comparison, in the product default (split-bf16 products) and in exact fp32: log-probs within 1e-3,
    decoding order and arg-max sequence identical on every unmasked residue."""
    dev = torch.device("cuda:0")
    cx = synth.make_complex(seed=77, n=1000, n_chains=4)
    fd = fd_of(cx, dev)
    m = make_model(weights_np, 48, dev)
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    ref = cpu_ref.score(w, fdc, 48)
    valid = torch.from_numpy(cx["mask"].astype(bool))
    for prec in ("x3", "fp32"):
        m.message_precision = prec
        sc = m.score(fd)
        assert torch.equal(sc["decoding_order"].cpu(), ref["decoding_order"])
        d = maxdiff(sc["log_probs"][0][valid], ref["log_probs"][0][valid])
        print(f"score() from coordinates, N = 1000, {prec}: max|dlogp| vs the CPU oracle = {d:.2e}")
        assert d < 1e-3, (prec, d)
        assert torch.equal(sc["log_probs"][0].argmax(-1).cpu()[valid], ref["log_probs"][0].argmax(-1)[valid]), prec


@pytest.mark.parametrize("prec", ["x3", "fp32"])
def test_ctor_variants_decode_protein_first_and_ref_atom(golden_dir, weights_np, prec):
    """Golden G4c (reference training copy built with decode_protein_first=1, na_model_utils.py:620-621, and with
    na_ref_atom="P", na_model_utils.py:497 / model_utils.py:573): the no-grad forward (inference kernels) and the
    differentiable forward (training kernels) of models built with the same constructor arguments."""
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "g4c_ctor_variants_n90_k24.npz"))
    n, k = 90, 24
    cx = synth.make_complex(seed=470, n=n, n_chains=4, masked_frac=0.03, missing_atom_frac=0.03)
    fd = fd_of(cx, dev)
    fd["S"] = fd["S"].long()
    valid = cx["mask"].astype(bool)
    randn = torch.from_numpy(g["randn"]).to(dev)

    def build(**kw):
        m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                        polytype_to_int=spec.polytype_to_int(), dropout=0.0, **kw)
        m.load_state_dict({k_: torch.from_numpy(v) for k_, v in weights_np.items()})
        m = m.to(dev).eval()
        m.message_precision = prec
        return m

    for tag, kw in (("dpf", dict(decode_protein_first=1)), ("refP", dict(na_ref_atom="P"))):
        m = build(**kw)
        ref = g[f"{tag}_log_probs"]
        with torch.no_grad():
            lp, p = m.forward(fd, decoding_randn=randn)
        assert maxdiff(lp[0], ref) < 1e-3, tag
        assert np.array_equal(lp[0].argmax(-1).cpu().numpy()[valid], ref.argmax(-1)[valid]), tag
        assert maxdiff(p[0].log()[valid], torch.from_numpy(ref)[valid]) < 1e-3
        with torch.enable_grad():                                   # the training path (na_mpnn_amd/train.py) on the same model
            lp_t, _ = m.forward(fd, decoding_randn=randn)
        assert lp_t.requires_grad
        assert maxdiff(lp_t[0].detach(), ref) < 1e-3, tag
        assert np.array_equal(lp_t[0].argmax(-1).cpu().numpy()[valid], ref.argmax(-1)[valid]), tag
        if tag == "dpf":
            chain_M = fd["mask"].masked_fill(fd["protein_mask"].to(torch.bool), 0)
            order = m.decoding_order(chain_M, randn)
            assert np.array_equal(order[0].cpu().numpy(), g["dpf_decoding_order"])
        else:
            V, E, E_idx = m.featurize(fd)
            ours_idx, ref_idx = E_idx[0].cpu().numpy(), g["refP_E_idx"].astype(np.int64)
            assert np.array_equal(np.sort(ours_idx, -1)[valid], np.sort(ref_idx, -1)[valid])
            compared = 0
            for q, i in enumerate(np.arange(n)[::max(1, n // 8)][:8]):
                if not valid[i]:
                    continue
                pos = {int(j): c for c, j in enumerate(ours_idx[i])}
                perm = torch.tensor([pos[int(j)] for j in ref_idx[i]], device=dev)
                assert maxdiff(E[0, i][perm], g["refP_E_rows"][q]) < 2e-4
                compared += 1
            assert compared >= 4
    # the default model on the same inputs differs from both variants (the arguments are not silently ignored)
    m0 = build()
    with torch.no_grad():
        lp0, _ = m0.forward(fd, decoding_randn=randn)
    assert maxdiff(lp0[0], g["dpf_log_probs"]) > 1e-3 and maxdiff(lp0[0], g["refP_log_probs"]) > 1e-3


@pytest.mark.parametrize("prec", ["x3", "fp32"])
def test_include_pred_na_N_0(golden_dir, prec):
    """A model built with include_pred_na_N=0 (na_model_utils.py:404-407,479-491: no virtual N_na atom, edge embedding
    [128 x 4640]) against the reference golden G4b: neighbour sets, edge-feature rows, and the training copy's forward."""
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "g4b_noN_n60_k24.npz"))
    w = synth.make_weights_noN(0)
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=24, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                    polytype_to_int=spec.polytype_to_int(), include_pred_na_N=0)
    assert tuple(m.features.edge_embedding.weight.shape) == (128, 4640)
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()})
    m = m.to(dev).eval()
    m.message_precision = prec
    n = 60
    cx = synth.make_complex(seed=460, n=n, n_chains=3, masked_frac=0.04, missing_atom_frac=0.03)
    fd = fd_of(cx, dev)
    V, E, E_idx = m.featurize(fd)
    valid = cx["mask"].astype(bool)
    ours_idx, ref_idx = E_idx[0].cpu().numpy(), g["E_idx"].astype(np.int64)
    assert np.array_equal(np.sort(ours_idx, -1)[valid], np.sort(ref_idx, -1)[valid])
    compared = 0
    for q, i in enumerate(np.arange(n)[::max(1, n // 8)][:8]):
        if not valid[i]:
            continue
        pos = {int(j): c for c, j in enumerate(ours_idx[i])}
        perm = torch.tensor([pos[int(j)] for j in ref_idx[i]], device=dev)
        assert maxdiff(E[0, i][perm], g["E_rows"][q]) < 2e-4
        compared += 1
    assert compared >= 4
    # the stock-ops featuriser takes the same expanded weight
    Vt, Et, It = featurize_torch(m, fd)
    if torch.equal(It[0][valid], E_idx[0][valid]):
        assert maxdiff(E[0][valid], Et[0][valid].cpu()) < 2e-4
    lp, p = m.forward(fd, decoding_randn=torch.from_numpy(g["randn"]).to(dev))
    assert maxdiff(lp[0], g["log_probs"]) < 1e-3
    assert np.array_equal(lp[0].argmax(-1).cpu().numpy()[valid], g["log_probs"].argmax(-1)[valid])


def test_score_batch_size_gt1_and_repack(weights_np):
    dev = torch.device("cuda:0")
    cx = synth.make_complex(seed=77, n=120)
    fd = fd_of(cx, dev)
    fd["batch_size"] = 4
    fd["randn"] = torch.randn(4, 120, device=dev)
    m = make_model(weights_np, 32, dev)
    out = m.score(fd)
    w = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    fdc = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in fd.items()}
    ref = cpu_ref.score(w, fdc, 32)
    assert out["log_probs"].shape == (4, 120, 33)
    assert maxdiff(out["log_probs"], ref["log_probs"]) < 1e-3
    assert torch.equal(out["decoding_order"].cpu(), ref["decoding_order"])
    # new weights -> automatic re-pack
    w2 = synth.make_weights(5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()})
    out2 = m.score(fd)
    ref2 = cpu_ref.score({k: torch.from_numpy(v) for k, v in w2.items()}, fdc, 32)
    assert maxdiff(out2["log_probs"], ref2["log_probs"]) < 1e-3


def _sample_fd(cx, dev, bs, T, randn):
    fd = fd_of(cx, dev)
    n = cx["S"].shape[0]
    fd.update({"batch_size": bs, "temperature": T, "bias": torch.zeros(1, n, 33, device=dev),
               "symmetry_residues": [[]], "symmetry_weights": [[]], "randn": randn.to(dev)})
    return fd


def test_sample_teacher_forced_golden(golden_dir, weights_np):
    """a9: sample() with the reference's own draws forced reproduces the reference's log_probs / probabilities (G5)."""
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "g5_sample.npz"))
    n, k, bs = 60, 16, 3
    cx = synth.make_complex(seed=500, n=n, n_chains=2)
    cx["chain_mask"][:7] = 0
    fd = _sample_fd(cx, dev, bs, 0.5, torch.from_numpy(g["randn"]))
    fd["S_forced"] = torch.from_numpy(g["S"].astype(np.int64)).to(dev)
    m = make_model(weights_np, k, dev)
    out = m.sample(fd)
    assert np.array_equal(out["decoding_order"].cpu().numpy(), g["decoding_order"])
    assert np.array_equal(out["S"].cpu().numpy(), g["S"].astype(np.int64))
    assert maxdiff(out["log_probs"], g["log_probs"]) < 1e-3
    assert maxdiff(out["sampling_probs"], g["sampling_probs"]) < 1e-3


def test_sample_symmetric_and_pair_bias_golden(golden_dir, weights_np):
    """Symmetry-tied groups (model_utils.py:219-326) and pair_bias (:169-172) with the reference's draws forced (G6)."""
    from test_oracle_golden import g6_inputs
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "g6_sample_variants.npz"))
    cx, k, fds, fdp = g6_inputs(g)
    m = make_model(weights_np, k, dev)
    to_dev = lambda fd: {k_: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    fds, fdp = to_dev(fds), to_dev(fdp)
    fds["S_forced"] = torch.from_numpy(g["sym_S"].astype(np.int64)).to(dev)
    out = m.sample(fds)
    assert np.array_equal(out["decoding_order"].cpu().numpy(), g["sym_order"])
    assert np.array_equal(out["S"].cpu().numpy(), g["sym_S"].astype(np.int64))
    assert maxdiff(out["log_probs"], g["sym_log_probs"]) < 1e-3
    assert maxdiff(out["sampling_probs"], g["sym_probs"]) < 1e-3
    fdp["S_forced"] = torch.from_numpy(g["pb_S"].astype(np.int64)).to(dev)
    out = m.sample(fdp)
    assert np.array_equal(out["S"].cpu().numpy(), g["pb_S"].astype(np.int64))
    assert maxdiff(out["log_probs"], g["pb_log_probs"]) < 1e-3
    assert maxdiff(out["sampling_probs"], g["pb_probs"]) < 1e-3
    # both together (golden G6b): the bias row of a group's LAST member, undrawn members read as PAD
    gb = np.load(os.path.join(golden_dir, "g6b_symmetric_pair_bias.npz"))
    fdq = dict(fds); fdq["pair_bias"] = fdp["pair_bias"]
    fdq["S_forced"] = torch.from_numpy(gb["S"].astype(np.int64)).to(dev)
    for lvl in (True, False):
        m.sample_level_parallel = lvl
        out = m.sample(fdq)
        assert np.array_equal(out["decoding_order"].cpu().numpy(), gb["order"])
        assert np.array_equal(out["S"].cpu().numpy(), gb["S"].astype(np.int64))
        assert maxdiff(out["log_probs"], gb["log_probs"]) < 1e-3
        assert maxdiff(out["sampling_probs"], gb["probs"]) < 1e-3


@pytest.mark.parametrize("variant", ["symmetric", "pair_bias"])
def test_sample_variants_free_running(weights_np, variant):
    """Free-running tied / pair-biased sampling: tied designable residues share one token, and the oracle teacher-forced
    with the sampled sequence reproduces log_probs and sampling probabilities."""
    from na_mpnn_amd.cli import make_pair_bias
    dev = torch.device("cuda:0")
    n, k, bs = 64, 24, 4
    cx = synth.make_complex(seed=777, n=n, n_chains=2)
    cx["chain_mask"][[2, 40]] = 0
    rng = np.random.default_rng(9)
    fd = _sample_fd(cx, dev, bs, 0.8, torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)))
    groups, gw = [[1, 2, 3], [10, 50], [40, 41], [20, 21, 22, 23, 24]], [[1., 1., 1.], [0.5, 1.5], [1., 2.], [1., -.5, 1., 1., 1.]]
    if variant == "symmetric":
        fd.update({"symmetry_residues": groups, "symmetry_weights": gw})
    else:
        fd["pair_bias"] = make_pair_bias(fd["chain_labels"][0], fd["R_idx"][0],
                                         torch.from_numpy(2.0 * rng.standard_normal((33, 33)).astype(np.float32)).to(dev))
    m = make_model(weights_np, k, dev)
    torch.manual_seed(3)
    out = m.sample(fd)
    S = out["S"].cpu()
    cm = (cx["mask"] * cx["chain_mask"]).astype(bool)
    assert torch.equal(S[:, ~cm], torch.from_numpy(cx["S"].astype(np.int64))[~cm].expand(bs, -1))
    for tok in (20, 25, 30, 31, 32):
        assert not (S[:, cm] == tok).any()
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    if variant == "symmetric":
        for grp in groups:      # one draw per group; a fixed member overrides the running token for the members after it
            for t_prev, t in zip(grp[:-1], grp[1:]):   # (the reference reuses S_t across the loop, model_utils.py:317-323)
                if cm[t]:
                    assert torch.equal(S[:, t], S[:, t_prev]), grp
        ref = cpu_ref.sample_symmetric(w, fdc, k, S_forced=S)
    else:
        ref = cpu_ref.sample(w, fdc, k, S_forced=S)
    assert torch.equal(ref["decoding_order"], out["decoding_order"].cpu())
    assert torch.equal(ref["S"], S)
    valid = torch.from_numpy(cx["mask"].astype(bool))
    assert maxdiff(out["log_probs"][:, valid], ref["log_probs"][:, valid]) < 1e-3
    assert maxdiff(out["sampling_probs"][:, valid], ref["sampling_probs"][:, valid]) < 1e-3


@pytest.mark.parametrize("n,k,bs,T,mf", [(70, 48, 5, 0.3, 0.0), (90, 32, 2, 1.0, 0.05), (50, 20, 1, 0.1, 0.05)])
def test_sample_free_running(weights_np, n, k, bs, T, mf):
    """Free-running sampler: (i) draws follow the returned distributions through the inverse CDF of the supplied
    uniforms, (ii) fixed positions keep S_true, special tokens never appear, (iii) the reference's stated
    invariant score(S_sampled).log_probs == sample().log_probs on designed positions (model_utils.py:367),
    (iv) teacher-forcing the oracle with the sampled S gives the same log_probs / probabilities."""
    dev = torch.device("cuda:0")
    cx = synth.make_complex(seed=600 + n, n=n, masked_frac=mf)
    cx["chain_mask"][::9] = 0
    # with masked AND fixed residues and batch_size > 1 the reference masks every stream with stream 0's mask
    # (model.reference_sample_mask_quirk) and thereby breaks its own score() invariant; check (iii) elsewhere
    check_score = (bs == 1) or (mf == 0.0)
    rng = np.random.default_rng(n)
    fd = _sample_fd(cx, dev, bs, T, torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)))
    m = make_model(weights_np, k, dev)
    torch.manual_seed(5)
    out = m.sample(fd)
    S, P, U = out["S"].cpu(), out["sampling_probs"].cpu(), out["uniform"].cpu()
    order = out["decoding_order"].cpu()
    cm = torch.from_numpy((cx["mask"] * cx["chain_mask"]).astype(bool))
    assert torch.equal(S[:, ~cm], torch.from_numpy(cx["S"].astype(np.int64))[~cm].expand(bs, -1))
    for tok in (20, 25, 30, 31, 32):
        assert not (S[:, cm] == tok).any()
    # (i) inverse CDF
    for b in range(bs):
        for t in range(n):
            i = int(order[b, t])
            if not cm[i]:
                continue
            cdf = torch.cumsum(P[b, i].double(), 0)
            u = float(U[b, t])
            expect = int((cdf > u).nonzero()[0]) if (cdf > u).any() else int(P[b, i].nonzero()[-1])
            if expect != int(S[b, i]):
                assert abs(float(cdf[min(expect, int(S[b, i]))]) - u) < 1e-5, (b, t, i)
    # (iii) score on the sampled sequence
    for b in range(bs if check_score else 0):
        fdb = dict(fd); fdb["batch_size"] = 1; fdb["S"] = S[b:b + 1].to(dev); fdb["randn"] = fd["randn"][b:b + 1]
        sc = m.score(fdb)
        d = (sc["log_probs"][0].cpu()[cm] - out["log_probs"][b].cpu()[cm]).abs().max()
        assert d < 2e-4, float(d)
    # (iv) oracle, teacher-forced
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    ref = cpu_ref.sample(w, fdc, k, S_forced=S)
    assert torch.equal(ref["decoding_order"], order)
    valid = torch.from_numpy(cx["mask"].astype(bool))
    assert maxdiff(out["log_probs"][:, valid], ref["log_probs"][:, valid]) < 1e-3
    assert maxdiff(out["sampling_probs"][:, valid], ref["sampling_probs"][:, valid]) < 1e-3


@pytest.mark.parametrize("prec", ["x3", "fp32"])
@pytest.mark.parametrize("n,k,bs,mf", [(120, 24, 3, 0.05), (300, 48, 1, 0.0), (40, 48, 2, 0.0), (97, 32, 1, 0.0), (700, 30, 4, 0.02)])
def test_level_parallel_sampling_equals_the_sequential_walk(weights_np, n, k, bs, mf, prec):
    """sample() decoded by dependency level — as ONE persistent launch walking the levels (round 3: device-side level offsets, grid
    barriers, no host read-back) and as one launch per level — gives bit-identical tokens, probabilities and log-probabilities to
    the one-launch sequential walk under the same uniforms, in both fp32-class precisions.  (700 x 4 visits = more work per level
    than the walk's grid holds in one pass.)"""
    dev = torch.device("cuda:0")
    cx = synth.make_complex(seed=900 + n, n=n, masked_frac=mf)
    cx["chain_mask"][::7] = 0
    rng = np.random.default_rng(n)
    fd = _sample_fd(cx, dev, bs, 0.4, torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)))
    m = make_model(weights_np, k, dev)
    m.message_precision = prec
    outs = []
    for lvl, walk in ((True, True), (True, False), (False, False)):
        m.sample_level_parallel, m.sample_level_walk = lvl, walk
        torch.manual_seed(21)
        outs.append(m.sample(fd))
        if lvl and walk:                                   # twice: the barrier words and the workspace are reused
            torch.manual_seed(21)
            again = m.sample(fd)
            assert torch.equal(again["S"], outs[-1]["S"]) and torch.equal(again["log_probs"], outs[-1]["log_probs"])
    w_, a, b = outs
    assert int(w_["levels"]) == a["levels"]
    assert "levels" in a and a["levels"] <= n and (a["levels"] < n or k >= n) and "levels" not in b   # complete graph: no parallelism
    for o in (w_, a):
        assert torch.equal(o["uniform"], b["uniform"]) and torch.equal(o["decoding_order"], b["decoding_order"])
        assert torch.isfinite(o["log_probs"]).all()
        assert torch.equal(o["S"], b["S"])
        assert torch.equal(o["sampling_probs"], b["sampling_probs"])
        assert torch.equal(o["log_probs"], b["log_probs"])


@pytest.mark.parametrize("kind", ["sequence_neighbours", "asymmetric", "dense"])
def test_pair_bias_sampling_by_level_equals_the_sequential_walk(weights_np, kind):
    """`pair_bias` (model_utils.py:116,169-172) decoded by dependency level: the residues whose tokens a step's bias reads become
    extra dependencies of the levels (namp_sample_levels_dep), and the level decoders treat every residue later in the decoding
    order as undecoded — tokens, probabilities and log-probabilities are bit-identical to the sequential walk.  "asymmetric": i
    reads j but j does not read i (j may sit in an earlier level although it comes later in the order); "dense": more than 64
    partners per residue keeps the sequential walk."""
    from na_mpnn_amd.cli import make_pair_bias
    dev = torch.device("cuda:0")
    n, k, bs = 80, 24, 3
    cx = synth.make_complex(seed=812, n=n, n_chains=2)
    cx["chain_mask"][[5, 33]] = 0
    rng = np.random.default_rng(12)
    fd = _sample_fd(cx, dev, bs, 0.7, torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)))
    AA = torch.from_numpy(2.0 * rng.standard_normal((33, 33)).astype(np.float32)).to(dev)
    if kind == "sequence_neighbours":
        pb = make_pair_bias(fd["chain_labels"][0], fd["R_idx"][0], AA)
    elif kind == "asymmetric":
        pb = torch.zeros(1, n, 33, n, 33, device=dev)
        for i in range(0, n - 9, 3):
            pb[0, i, :, i + 9, :] = AA                       # i reads i + 9; i + 9 does not read i
            pb[0, i + 2, :, (i * 7) % n, :] = AA.t()
    else:
        pb = 0.05 * torch.from_numpy(rng.standard_normal((1, n, 33, n, 33)).astype(np.float32)).to(dev)
    fd["pair_bias"] = pb
    m = make_model(weights_np, k, dev)
    outs = []
    for lvl, walk in ((True, True), (True, False), (False, False)):
        m.sample_level_parallel, m.sample_level_walk = lvl, walk
        torch.manual_seed(8)
        outs.append(m.sample(fd))
    w_, a, b = outs
    assert ("levels" in a) == (kind != "dense") and "levels" not in b
    if kind != "dense":
        assert int(w_["levels"]) == a["levels"] < n
    for o in (w_, a):
        assert torch.equal(o["S"], b["S"]) and torch.equal(o["sampling_probs"], b["sampling_probs"]) and torch.equal(o["log_probs"], b["log_probs"])
    # and the sequential walk itself is the reference's: teacher-forced oracle
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    ref = cpu_ref.sample(w, fdc, k, S_forced=b["S"].cpu())
    valid = torch.from_numpy(cx["mask"].astype(bool))
    assert maxdiff(w_["log_probs"][:, valid], ref["log_probs"][:, valid]) < 1e-3


@pytest.mark.parametrize("kind", ["mixed_groups", "homo_dimer", "with_pair_bias"])
def test_symmetric_sampling_by_level_equals_the_sequential_walk(weights_np, kind):
    """Symmetry-tied sampling (model_utils.py:219-326) decoded by dependency level (round 3): a work item is a whole group, its members run
    one after the other in one workgroup slot (a member gathers the decoder states of the members before it), the group's level is
    1 + the highest level of its members' dependencies in earlier groups.  Tokens, probabilities and log-probabilities are bit-identical
    to the sequential walk in both level forms — and with the groups whose members are not graph neighbours of each other split into
    single-member work items with a deferred draw (`sample_split_groups`); the walk itself is checked against the oracle's tied sampler.
    "mixed_groups": groups of 2-5 (one of consecutive residues, i.e. graph neighbours of each other; one with a fixed member; negative
    weights) among free residues; "homo_dimer": every residue tied to its copy in the other chain; "with_pair_bias": groups + pair_bias."""
    from na_mpnn_amd.cli import make_pair_bias
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    if kind == "homo_dimer":
        n, k, bs = 120, 32, 2
        cx = synth.make_complex(seed=901, n=n, n_chains=2)
        groups = [[i, i + 60] for i in range(60)]
        gw = [[0.5, 0.5] for _ in range(60)]
    else:
        n, k, bs = 96, 24, 3
        cx = synth.make_complex(seed=902, n=n, n_chains=2, masked_frac=0.03)
        cx["chain_mask"][[2, 40, 71]] = 0
        groups = [[1, 2, 3], [10, 50], [40, 41], [20, 21, 22, 23, 24], [60, 7, 90, 33], [70, 71]]
        gw = [[1., 1., 1.], [0.5, 1.5], [1., 2.], [1., -.5, 1., 1., 1.], [.25, .25, .25, .25], [1., 1.]]
    fd = _sample_fd(cx, dev, bs, 0.6, torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)))
    fd.update({"symmetry_residues": groups, "symmetry_weights": gw})
    if kind == "with_pair_bias":
        fd["pair_bias"] = make_pair_bias(fd["chain_labels"][0], fd["R_idx"][0],
                                         torch.from_numpy(2.0 * rng.standard_normal((33, 33)).astype(np.float32)).to(dev))
    m = make_model(weights_np, k, dev)
    outs = []
    for lvl, walk, split in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
        m.sample_level_parallel, m.sample_level_walk, m.sample_split_groups = lvl, walk, split
        torch.manual_seed(11)
        outs.append(m.sample(fd))
    w_, ws_, a, b = outs
    assert m.sample_walk_status() == 0
    assert "levels" in a and "levels" not in b and int(w_["levels"]) == a["levels"] == int(ws_["levels"])
    n_groups = n - sum(len(g_) - 1 for g_ in groups)
    assert a["levels"] < n_groups, (a["levels"], n_groups)
    # groups without internal graph edges were split into single-member items (more items than groups), the others kept whole
    assert ws_["work_items"] == bs * n_groups and bs * n_groups < w_["work_items"] <= bs * n, (w_["work_items"], n_groups)
    for o in (w_, ws_, a):
        assert torch.equal(o["decoding_order"], b["decoding_order"])
        assert torch.equal(o["S"], b["S"]) and torch.equal(o["sampling_probs"], b["sampling_probs"]) and torch.equal(o["log_probs"], b["log_probs"])
    S = b["S"].cpu()
    cm = (cx["mask"] * cx["chain_mask"]).astype(bool)
    for grp in groups:
        for t_prev, t in zip(grp[:-1], grp[1:]):
            if cm[t]:
                assert torch.equal(S[:, t], S[:, t_prev]), grp
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    ref = cpu_ref.sample_symmetric(w, fdc, k, S_forced=S)
    valid = torch.from_numpy(cx["mask"].astype(bool))
    assert torch.equal(ref["decoding_order"], w_["decoding_order"].cpu()) and torch.equal(ref["S"], S)
    assert maxdiff(w_["log_probs"][:, valid], ref["log_probs"][:, valid]) < 1e-3
    assert maxdiff(w_["sampling_probs"][:, valid], ref["sampling_probs"][:, valid]) < 1e-3


def test_sampler_with_more_than_three_decoder_layers():
    """The sampler kernels index their per-layer arguments at run time (round 3): a model with 5 decoder layers — more than the
    reference configuration's 3, up to NAMP_MAX_LAYERS = 8 — samples through all three forms (persistent level walk, one launch
    per level, sequential walk) with identical draws, and its teacher-forced log-probs match the oracle's step-by-step decoder."""
    dev = torch.device("cuda:0")
    n, k, bs = 60, 24, 2
    w_np = synth.make_weights(5, num_encoder_layers=2, num_decoder_layers=5)
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, num_encoder_layers=2, num_decoder_layers=5, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w_np.items()})
    m = m.to(dev).eval()
    cx = synth.make_complex(seed=555, n=n, masked_frac=0.05)
    rng = np.random.default_rng(5)
    fd = _sample_fd(cx, dev, bs, 0.5, torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)))
    outs = []
    for lvl, walk in ((True, True), (True, False), (False, False)):
        m.sample_level_parallel, m.sample_level_walk = lvl, walk
        torch.manual_seed(3)
        outs.append(m.sample(fd))
    for o in outs[:2]:
        assert torch.equal(o["S"], outs[2]["S"]) and torch.equal(o["log_probs"], outs[2]["log_probs"])
    S = outs[0]["S"].cpu()
    w = {k_: torch.from_numpy(v) for k_, v in w_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    ref = cpu_ref.sample(w, fdc, k, S_forced=S)
    valid = torch.from_numpy(cx["mask"].astype(bool))
    assert torch.equal(ref["S"], S)
    assert maxdiff(outs[0]["log_probs"][:, valid], ref["log_probs"][:, valid]) < 1e-3
    assert maxdiff(outs[0]["sampling_probs"][:, valid], ref["sampling_probs"][:, valid]) < 1e-3


def test_cpu_tensors_are_rejected(weights_np):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=8, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    cx = synth.make_complex(seed=1, n=20)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in cx.items()}
    fd["batch_size"] = 1
    with pytest.raises(RuntimeError, match="HIP device"):
        m.score(fd)


def test_cli_design_and_specificity(tmp_path, weights_np):
    """f3: run.py-compatible CLI end to end on a synthetic PDB (seeded synthetic weights): FASTA header/sequence
    format of run.py:445-455,501-511 and the specificity .npz keys of run.py:426-443."""
    from na_mpnn_amd import cli, pdbio
    cx = synth.make_complex(seed=31, n=64, n_chains=3)
    int_to_res = {v: k for k, v in spec.restype_to_int().items()}
    letters = ["ABC"[c] for c in cx["chain_labels"]]
    pdb = os.path.join(str(tmp_path), "toy.pdb")
    pdbio.write_pdb(pdb, cx["X"], cx["X_m"], [int_to_res[int(s)] for s in cx["S"]], letters, cx["R_idx"])
    out = os.path.join(str(tmp_path), "out")
    cli.main(["--mode", "design", "--pdb_path", pdb, "--out_folder", out, "--random_init_seed", "0", "--seed", "7",
              "--fixed_residues", f"A{cx['R_idx'][0]} A{cx['R_idx'][1]}", "--number_of_batches", "2"])
    lines = open(os.path.join(out, "seqs", "toy.fa")).read().splitlines()
    assert len(lines) == 2 * (1 + 2)
    assert lines[0].startswith(">toy, T=0.1, seed=7, num_res=62, batch_size=1, number_of_batches=2, model_path=")
    assert lines[2].startswith(">toy, id=1, T=0.1, seed=7, overall_confidence=") and " seq_rec=" in lines[2]
    native, designed = lines[1], lines[3]
    assert len(native) == 64 + 2 and native.count("/") == 2 and len(designed) == len(native)
    assert designed[:2] == native[:2]                       # the two fixed residues keep their identity
    assert not set(designed) & set("Xxy-+")                 # omitted / special tokens never appear
    cli.main(["--mode", "specificity", "--pdb_path", pdb, "--out_folder", out, "--random_init_seed", "0", "--seed", "7",
              "--design_na_only", "1", "--output_specificity", "1", "--batch_size", "6"])
    z = np.load(os.path.join(out, "specificity", "toy.npz"), allow_pickle=True)
    assert z["predicted_ppm"].shape == (64, 33)
    na = (cx["dna_mask"] + cx["rna_mask"]).astype(bool)
    na_chains = {l for l, f in zip(letters, na) if f}       # --design_na_only works per chain (run.py:296-301)
    des = np.array([l in na_chains for l in letters])
    assert np.allclose(z["predicted_ppm"][des].sum(-1), 1.0, atol=1e-5) and np.all(z["predicted_ppm"][~des] == 0)
    assert list(z["encoded_residues"][:2]) == [f"A{cx['R_idx'][0]}", f"A{cx['R_idx'][1]}"]


@pytest.mark.parametrize("ext", ["pdb", "cif"])
def test_cli_output_files_byte_level(tmp_path, golden_dir, ext):
    """f3: the design CLI's OUTPUT FILES against fixtures written with the reference's format strings from the CPU oracle's
    sample() (oracle/make_cli_fixture.py): with the oracle's draws forced (`--forced_draws_npz`), seqs/<name>.fa is
    byte-identical (headers of run.py:445-455 / :501-511, 4-digit confidence / recovery, chain-separated sequences with the
    RNA letter conversion), specificity/<name>.npz has the keys, dtypes and values of run.py:426-443, and backbones/ carries
    the designed residue names (run.py:475-491) — from the PDB and from the mmCIF form of the same complex."""
    from na_mpnn_amd import cli, pdbio
    gd = os.path.join(golden_dir, "cli")
    out = os.path.join(str(tmp_path), "out")
    cli.main(["--pdb_path", os.path.join(gd, "input." + ext), "--out_folder", out, "--random_init_seed", "0", "--seed", "7",
              "--batch_size", "2", "--temperature", "1.0", "--fixed_residues", "A0 A1", "--output_specificity", "1",
              "--forced_draws_npz", os.path.join(gd, "forced_draws.npz")])
    got = open(os.path.join(out, "seqs", "input.fa"), "rb").read()
    want = open(os.path.join(gd, "expected.fa"), "rb").read()
    assert got == want, (got.decode(), want.decode())
    z = np.load(os.path.join(out, "specificity", "input.npz"), allow_pickle=True)
    e = np.load(os.path.join(gd, "expected_specificity.npz"), allow_pickle=True)
    assert sorted(z.files) == sorted(e.files)
    for k in e.files:
        assert z[k].dtype == e[k].dtype and z[k].shape == e[k].shape, k
        if k == "predicted_ppm":
            assert np.abs(z[k] - e[k]).max() < 1e-4
        elif z[k].dtype == object:
            assert z[k].item() == e[k].item(), k
        else:
            assert np.array_equal(z[k], e[k]), k
    forced = np.load(os.path.join(gd, "forced_draws.npz"))
    P = pdbio.parse_pdb(os.path.join(gd, "input.pdb"))
    rti = spec.restype_to_int(True)
    int_to_3 = {}
    for k3, v in rti.items():
        int_to_3.setdefault(v, k3)
    dna_to_rna3 = {"DA": "A", "DC": "C", "DG": "G", "DT": "U", "DX": "RX"}
    for ix in (1, 2):
        lines = open(os.path.join(out, "backbones", f"input_{ix}.pdb")).read().splitlines()
        atoms = [l for l in lines if l.startswith("ATOM")]
        assert len(atoms) == int(P["X_m"].sum()) and lines[-1] == "END"
        assert sum(l.startswith("HETATM") for l in lines) == (1 if ext == "pdb" else 0)   # the ligand travels; waters do not
        # every backbone atom keeps its coordinates and carries the DESIGNED residue name (a random-init model may put any
        # letter anywhere, so the file is checked line by line rather than re-parsed)
        want_name = {}
        for i, (c, r) in enumerate(zip(P["chain_letters"], P["R_idx"].tolist())):
            n3 = int_to_3[int(forced["S_forced"][ix - 1][i])]
            if P["rna_mask_for_token_conversion"][i] == 1:
                n3 = dna_to_rna3.get(n3, n3)
            want_name[(c, r)] = n3
        xyz = {(P["chain_letters"][i], int(P["R_idx"][i]), a): P["X"][i, j] for i in range(len(P["S"]))
               for j, a in enumerate(spec.ATOM_TYPES) if P["X_m"][i, j]}
        for l in atoms:
            key = (l[21], int(l[22:26]))
            assert l[17:20].strip() == want_name[key], l
            got_xyz = np.array([float(l[30:38]), float(l[38:46]), float(l[46:54])])
            assert np.abs(got_xyz - xyz[key + (l[12:16].strip(),)]).max() < 1e-3


def test_padded_batch_from_coordinates(weights_np):
    """G8: B=3 complexes of different length padded with mask=0 tails, through the drop-in surface's training-style
    forward (B > 1): every real residue must match the same complex run alone (padding invariance) and the oracle."""
    dev = torch.device("cuda:0")
    ns, Lmax, k = [70, 120, 95], 120, 32
    cxs = [synth.make_complex(seed=800 + i, n=n) for i, n in enumerate(ns)]
    pad = {}
    for key in cxs[0]:
        arrs = []
        for cx in cxs:
            a = cx[key]
            p = np.zeros((Lmax - a.shape[0],) + a.shape[1:], a.dtype)
            if key == "R_polymer_type":
                p += 5                                    # PAD polymer type
            if key == "R_idx":
                p -= 100
            if key == "chain_labels":
                p -= 1
            arrs.append(np.concatenate([a, p], 0))
        pad[key] = np.stack(arrs)
    fd = {k_: torch.from_numpy(v).to(dev) for k_, v in pad.items()}
    m = make_model(weights_np, k, dev)
    randn = torch.randn(3, Lmax, device=dev)
    lp, _ = m.forward(fd, decoding_randn=randn)
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: v.cpu() for k_, v in fd.items()}
    fdc["S"] = fdc["S"].long()
    ref, _ = cpu_ref.forward_train(w, fdc, k, randn.cpu())
    for b, n in enumerate(ns):
        assert maxdiff(lp[b, :n], ref[b, :n]) < 1e-3
        assert torch.equal(lp[b, :n].argmax(-1).cpu(), ref[b, :n].argmax(-1))


@pytest.mark.parametrize("n,k,bs", [(300, 100, 1), (200, 130, 1), (1000, 64, 1), (1000, 65, 2), (5000, 48, 1), (8192, 48, 1),
                                    (700, 30, 3), (129, 33, 1)])
def test_knn_selection_equals_the_full_row_sort(weights_np, n, k, bs, monkeypatch):
    """knn_select_kernel (radix select + small sort) against knn_kernel (bitonic sort of the whole row) on the same keys:
    identical neighbour lists in identical order — with coincident residues (equal distances: ties go to the lower index),
    masked residues (their substitute distance ties with the row maximum) and K from 30 to 130 (final sorts of 64, 128, 256
    keys; K close to L falls back to the full sort on both sides)."""
    dev = torch.device("cuda:0")
    cxs = []
    for b in range(bs):
        cx = synth.make_complex(seed=900 + n + b, n=n, masked_frac=0.1)
        dup = np.random.default_rng(n + b).integers(0, n, size=max(1, n // 10))
        cx["X"][dup] = cx["X"][(dup + 7) % n]                       # coincident residues
        cxs.append(cx)
    from na_mpnn_amd import shard
    fd = {k_: v.to(dev) for k_, v in shard.pad_batch(cxs).items()} if bs > 1 else fd_of(cxs[0], dev)
    fd["batch_size"] = 1
    m = make_model(weights_np, k, dev)
    monkeypatch.delenv("NAMP_KNN_FULL_SORT", raising=False)
    sel = m.featurize(fd)[2].clone()
    monkeypatch.setenv("NAMP_KNN_FULL_SORT", "1")
    full = m.featurize(fd)[2].clone()
    assert sel.shape[-1] == min(k, n)
    assert torch.equal(sel, full)


@pytest.mark.parametrize("n,k,kw", [(1, 48, {}), (2, 48, {}), (3, 1, {}), (17, 16, {}), (40, 17, dict(masked_frac=0.3)),
                                    (33, 5, dict(missing_atom_frac=0.3)), (64, 64, {}), (65, 63, dict(n_chains=7)),
                                    (129, 48, dict(frac_protein=0.0, frac_dna=1.0)), (50, 48, dict(frac_protein=1.0, frac_dna=0.0)),
                                    (257, 100, {}), (90, 150, dict(masked_frac=0.5))])
def test_extreme_shapes_score_and_sample(weights_np, n, k, kw):
    """Edge shapes through the whole drop-in surface: single residues, K = 1, K >= L, K > 64, heavy masking, single-polymer
    complexes.  (Masked cases keep K below the number of unmasked residues or at L: in between, top-k must choose among
    masked residues that all tie at the row maximum, which torch.topk leaves unspecified — the reference itself is
    ambiguous there.)  score() / unconditional_probs() against the oracle from coordinates; sample() against the teacher-forced
    oracle and against its own sequential walk."""
    dev = torch.device("cuda:0")
    cx = synth.make_complex(seed=5000 + 7 * n + k, n=n, **{"n_chains": min(3, n), **kw})
    if n > 4:
        cx["chain_mask"][::5] = 0
    fd = fd_of(cx, dev)
    rng = np.random.default_rng(n * 1000 + k)
    fd["randn"] = torch.from_numpy(rng.standard_normal((1, n)).astype(np.float32)).to(dev)
    m = make_model(weights_np, k, dev)
    w = {k_: torch.from_numpy(v) for k_, v in weights_np.items()}
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    valid = torch.from_numpy(cx["mask"].astype(bool))
    sc, ref = m.score(fd), cpu_ref.score(w, fdc, k)
    assert torch.equal(sc["decoding_order"].cpu(), ref["decoding_order"])
    if valid.any():
        assert maxdiff(sc["log_probs"][0][valid], ref["log_probs"][0][valid]) < 1e-3
        assert maxdiff(m.unconditional_probs(fd)["log_probs"][0][valid],
                       cpu_ref.unconditional_probs(w, fdc, k)["log_probs"][0][valid]) < 1e-3
    assert torch.isfinite(sc["log_probs"]).all()
    # sampler: level-parallel == sequential, and both == oracle with the drawn sequence forced
    bs = 2
    fd.update({"batch_size": bs, "temperature": 0.5, "bias": torch.zeros(1, n, 33, device=dev),
               "symmetry_residues": [[]], "symmetry_weights": [[]],
               "randn": torch.from_numpy(rng.standard_normal((bs, n)).astype(np.float32)).to(dev)})
    outs = []
    for lvl in (True, False):
        m.sample_level_parallel = lvl
        torch.manual_seed(n)
        outs.append(m.sample(fd))
    assert torch.equal(outs[0]["S"], outs[1]["S"]) and torch.equal(outs[0]["log_probs"], outs[1]["log_probs"])
    assert torch.equal(outs[0]["sampling_probs"], outs[1]["sampling_probs"])
    fdc = {k_: (v.cpu() if isinstance(v, torch.Tensor) else v) for k_, v in fd.items()}
    # masked + fixed residues with batch_size > 1: the reference masks all streams with stream 0's mask (quirk kept)
    refs = cpu_ref.sample(w, fdc, k, S_forced=outs[0]["S"].cpu())
    if valid.any():
        assert maxdiff(outs[0]["log_probs"][:, valid], refs["log_probs"][:, valid]) < 1e-3
        assert maxdiff(outs[0]["sampling_probs"][:, valid], refs["sampling_probs"][:, valid]) < 1e-3


def test_maximum_size_complex_padding_invariance(weights_np):
    """N = 6000 (the cap of the design_test-sized split) from coordinates: rows are normalised and finite, and
    every residue's log-probs are unchanged when the complex sits in a padded batch next to a shorter one (size-independent
    property; the oracle would need ~2 GB of RBF temporaries here)."""
    from na_mpnn_amd import shard
    dev = torch.device("cuda:0")
    big = synth.make_complex(seed=6000, n=6000, n_chains=6)
    small = synth.make_complex(seed=6001, n=777, n_chains=2)
    m = make_model(weights_np, 48, dev)
    randn = torch.randn(2, 6000, generator=torch.Generator().manual_seed(1)).to(dev)
    fd1 = fd_of(big, dev); fd1["randn"] = randn[:1]
    lp1 = m.score(fd1)["log_probs"]
    assert torch.isfinite(lp1).all()
    assert float((torch.logsumexp(lp1, -1)).abs().max()) < 1e-4
    fd2 = shard.pad_batch([big, small], device=dev)
    fd2["batch_size"] = 1; fd2["randn"] = randn
    lp2 = m.score(fd2)["log_probs"]
    assert float((lp2[0] - lp1[0]).abs().max()) < 1e-4          # different launch shapes (fused vs unfused tails), same rows
    assert torch.equal(lp2[0].argmax(-1), lp1[0].argmax(-1))
    fd3 = fd_of(small, dev); fd3["randn"] = randn[1:, :777]
    lp3 = m.score(fd3)["log_probs"]
    assert float((lp2[1, :777] - lp3[0]).abs().max()) < 1e-4

