"""Deterministic synthetic weights and inputs (numpy ``default_rng`` only).

SURVEY §7 step 1 / §8(d): the trained checkpoints are not in the reference tree
(`.MISSING_LARGE_BLOBS`), so every golden, parity test and bench run regenerates
the same seeded tensors on whichever box it runs — no 9 MB weight file ships.
Nothing here touches torch RNG; conversion to torch happens at the call site.
"""
from __future__ import annotations

import numpy as np

from . import spec


def make_weights(seed: int = 0, num_encoder_layers: int = 3, num_decoder_layers: int = 3):
    """{state_dict key: float32 ndarray}.  Matrices are Xavier-uniform (what the
    reference constructors apply, model_utils.py:67-69); biases and LayerNorm
    affine terms are drawn non-trivially so that every term is exercised."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in spec.state_dict_spec(num_encoder_layers, num_decoder_layers).items():
        if len(shape) == 2:
            fan_out, fan_in = shape
            a = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-a, a, size=shape)
        elif ".norm" in key or "norm_" in key:
            w = (1.0 + 0.1 * rng.standard_normal(shape)) if key.endswith("weight") \
                else 0.1 * rng.standard_normal(shape)
        else:  # Linear bias
            w = 0.1 * rng.standard_normal(shape)
        out[key] = w.astype(np.float32)
    return out


def make_weights_noN(seed: int = 0):
    """make_weights() for a model built with include_pred_na_N=0 (na_model_utils.py:404-407): the edge embedding sees
    17 x 17 atom pairs, i.e. features.edge_embedding.weight is [128 x 4640]; everything else is unchanged."""
    w = make_weights(seed)
    n_in = spec.NUM_POS + spec.NUM_RBF * 17 * 17
    a = np.sqrt(6.0 / (n_in + spec.H))
    w["features.edge_embedding.weight"] = np.random.default_rng(seed + 4640).uniform(-a, a, size=(spec.H, n_in)).astype(np.float32)
    return w


def random_walk_backbone(rng, n: int, step: float = 3.8):
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.cumsum(step * d, axis=0)


def make_complex(seed: int, n: int, *, n_chains: int = 4, frac_protein: float = 0.70,
                 frac_dna: float = 0.15, masked_frac: float = 0.0, missing_atom_frac: float = 0.0):
    """One synthetic complex as the reference ``feature_dict`` (no batch dim).

    Layout / dtypes follow inference/data_utils.py:361-394 (inference) — X f32
    [L,16,3], X_m i32 [L,16], mask i32, S i32, R_idx i32, chain_labels i32,
    protein/dna/rna_mask i32, R_polymer_type i64.
    """
    rng = np.random.default_rng(seed)
    centre = random_walk_backbone(rng, n)
    X = (centre[:, None, :] + 1.5 * rng.standard_normal((n, spec.N_ATOMS, 3))).astype(np.float32)

    # contiguous polymer blocks split in chains
    n_prot = int(round(frac_protein * n))
    n_dna = int(round(frac_dna * n))
    n_rna = n - n_prot - n_dna
    poly = np.concatenate([np.zeros(n_prot, np.int64), np.ones(n_dna, np.int64),
                           2 * np.ones(n_rna, np.int64)])
    bounds = np.sort(rng.choice(np.arange(1, n), size=min(n_chains - 1, n - 1), replace=False)) \
        if n > 1 and n_chains > 1 else np.array([], dtype=np.int64)
    chain_labels = np.searchsorted(bounds, np.arange(n), side="right").astype(np.int32)
    R_idx = np.arange(n, dtype=np.int32)
    # residue numbering restarts per chain with a +100 jump like parsed PDBs
    for c in range(int(chain_labels.max()) + 1):
        sel = chain_labels == c
        R_idx[sel] = np.arange(sel.sum(), dtype=np.int32) + 100 * c

    protein_mask = (poly == 0).astype(np.int32)
    dna_mask = (poly == 1).astype(np.int32)
    rna_mask = (poly == 2).astype(np.int32)
    X_m = np.zeros((n, spec.N_ATOMS), np.int32)
    X_m[poly == 0, :4] = 1
    X_m[poly != 0, 4:] = 1
    X_m[poly == 1, 14] = 0           # DNA has no O2' (that is how the reference's parser tells DNA from RNA)
    if missing_atom_frac > 0:
        # never the kNN reference atoms CA / C1': the reference's parser masks such residues out
        # (data_utils.py), and zeroed reference points would create exact distance ties in topk
        drop = rng.random((n, spec.N_ATOMS)) < missing_atom_frac
        drop[:, [1, 15]] = False
        X_m[drop] = 0
    X = (X * X_m[:, :, None]).astype(np.float32)   # absent atoms are stored as 0 (data_utils.py zero-fills)

    S = np.where(poly == 0, rng.integers(0, 20, n),
                 np.where(poly == 1, rng.integers(21, 25, n), rng.integers(26, 30, n))).astype(np.int32)
    mask = np.ones(n, np.int32)
    if masked_frac > 0:
        mask[rng.random(n) < masked_frac] = 0
    return {
        "X": X, "X_m": X_m, "mask": mask, "S": S, "R_idx": R_idx, "chain_labels": chain_labels,
        "protein_mask": protein_mask, "dna_mask": dna_mask, "rna_mask": rna_mask,
        "R_polymer_type": poly, "chain_mask": np.ones(n, np.int32),
        "randn": rng.standard_normal(n).astype(np.float32),
    }


def knn_indices(centre: np.ndarray, k: int) -> np.ndarray:
    """Brute-force k nearest neighbours (self first) of [n,3] points -> int32 [n,k]."""
    d2 = ((centre[:, None, :] - centre[None, :, :]) ** 2).sum(-1)
    k = min(k, centre.shape[0])
    idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return idx.astype(np.int32)


def _layer_norm_rows(x):
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + 1e-5)


def make_graph(seed: int, batch: int, n: int, k: int, *, masked_frac: float = 0.0):
    """Synthetic encoder/decoder inputs *after* featurisation:
    V,E = LayerNorm-ed randn (the statistics of the reference features output),
    E_idx = true kNN of a random walk, S uniform over the 25 non-special tokens.
    Returns numpy arrays with a leading batch dimension."""
    rng = np.random.default_rng(seed)
    k_eff = min(k, n)
    V = np.empty((batch, n, spec.H), np.float32)
    E = np.empty((batch, n, k_eff, spec.H), np.float32)
    E_idx = np.empty((batch, n, k_eff), np.int32)
    for b in range(batch):
        centre = random_walk_backbone(rng, n)
        E_idx[b] = knn_indices(centre, k_eff)
        V[b] = _layer_norm_rows(rng.standard_normal((n, spec.H))).astype(np.float32)
        E[b] = _layer_norm_rows(rng.standard_normal((n, k_eff, spec.H), dtype=np.float32))
    S = rng.integers(0, 25, (batch, n)).astype(np.int32)
    mask = np.ones((batch, n), np.int32)
    if masked_frac > 0:
        mask[rng.random((batch, n)) < masked_frac] = 0
    chain_mask = np.ones((batch, n), np.int32)
    randn = rng.standard_normal((batch, n)).astype(np.float32)
    return {"V": V, "E": E, "E_idx": E_idx, "S": S, "mask": mask,
            "chain_mask": chain_mask, "randn": randn}
