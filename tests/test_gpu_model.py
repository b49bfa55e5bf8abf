"""GPU tests of the drop-in Python surface (na_mpnn_amd.model.ProteinMPNN) against the goldens the
real reference produced from coordinates (G4/G6) and against the oracle."""
import os

import numpy as np
import pytest
import torch

from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
from oracle import cpu_ref

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


@pytest.mark.parametrize("n,k,tag,kw", [(97, 32, "n97_k32", dict(missing_atom_frac=0.05, masked_frac=0.04)),
                                         (150, 48, "n150_k48", {}), (32, 48, "n32_k48_LltK", {})])
def test_score_from_coordinates(golden_dir, weights_np, n, k, tag, kw):
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, f"g4_fromX_{tag}.npz"))
    cx = synth.make_complex(seed=400 + n, n=n, **kw)
    fd = fd_of(cx, dev)
    m = make_model(weights_np, k, dev)
    V, E, E_idx = m.featurize(fd)
    # neighbour sets of every unmasked residue must agree with the reference.  (A masked residue's row of
    # the distance matrix is all-equal, so torch.topk's pick there is an arbitrary tie-break on either
    # device — model_utils.py:493-496 — and it cannot influence any unmasked output.)
    valid = cx["mask"].astype(bool)
    ref_idx = np.sort(g["E_idx"].astype(np.int64), -1)
    assert np.array_equal(np.sort(E_idx[0].cpu().numpy(), -1)[valid], ref_idx[valid])
    assert maxdiff(V[0], g["V"]) < 1e-5
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


def test_cpu_tensors_are_rejected(weights_np):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=8, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    cx = synth.make_complex(seed=1, n=20)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in cx.items()}
    fd["batch_size"] = 1
    with pytest.raises(RuntimeError, match="HIP device"):
        m.score(fd)
