"""Static vocabulary / shape contract of the NA-MPNN hot path.

These tables are *data* the reference hard-codes inside its CLI
(/root/reference/inference/run.py:15-66 atom / polymer / residue vocabularies,
:112-117 the shared DNA/RNA token aliasing) and that the model constructors take
as ``atom_dict`` / ``polytype_to_int`` / ``restype_to_int``.  They are restated
here so that the drop-in module, the oracle and the tests agree on one copy.
"""
from __future__ import annotations

from collections import OrderedDict

H = 128            # hidden / node / edge feature width
FFN = 4 * H        # PositionWiseFeedForward inner width (model_utils.py:634)
VOCAB = 33         # num_letters == vocab (run.py:130-131)
NUM_RBF = 16
NUM_POS = 16
MAX_REL = 32       # PositionalEncodings.max_relative_feature (model_utils.py:607)
NUM_POS_CLASSES = 2 * MAX_REL + 2   # relative offsets -32..32 within a chain + one class for other chains
MSG_SCALE = 30.0   # EncLayer/DecLayer ``scale`` (model_utils.py:620,660)
LN_EPS = 1e-5

ATOM_TYPES = ["N", "CA", "C", "O",
              "OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "O2'", "C1'"]
POLYTYPES = ["PP", "DNA", "RNA", "UNK", "MAS", "PAD"]
RESTYPES = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
            "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL", "UNK",
            "DA", "DC", "DG", "DT", "DX", "A", "C", "G", "U", "RX", "MAS", "PAD"]
RESTYPE_3TO1 = dict(zip(RESTYPES, list("ARNDCQEGHILKMFPSTWYVXacgtxbdhuy-+")))

N_ATOMS = len(ATOM_TYPES)          # 16
N_ATOMS_AUG = N_ATOMS + 2          # + virtual Cb + virtual N_na (model_utils.py:478-482)
EDGE_IN = NUM_POS + NUM_RBF * N_ATOMS_AUG * N_ATOMS_AUG   # 5200


def atom_dict():
    return dict(zip(ATOM_TYPES, range(N_ATOMS)))


def polytype_to_int():
    return dict(zip(POLYTYPES, range(len(POLYTYPES))))


def restype_to_int(na_shared_tokens: bool = False):
    d = dict(zip(RESTYPES, range(len(RESTYPES))))
    if na_shared_tokens:   # run.py:112-117
        d["A"], d["C"], d["G"], d["U"], d["RX"] = d["DA"], d["DC"], d["DG"], d["DT"], d["DX"]
    return d


def state_dict_spec(num_encoder_layers: int = 3, num_decoder_layers: int = 3,
                    hidden: int = H, vocab: int = VOCAB, num_letters: int = VOCAB):
    """Ordered {key: shape} of the reference ``ProteinMPNN.state_dict()``.

    Same key set for the inference copy (inference/model_utils.py:8-69) and the
    training copy (na_model_utils.py:519-587); SURVEY App. A.6.
    """
    s = OrderedDict()
    h = hidden

    def lin(name, out_f, in_f, bias=True):
        s[name + ".weight"] = (out_f, in_f)
        if bias:
            s[name + ".bias"] = (out_f,)

    def ln(name):
        s[name + ".weight"] = (h,)
        s[name + ".bias"] = (h,)

    lin("W_v", h, h)
    lin("features.embeddings.linear", NUM_POS, 2 * MAX_REL + 2)
    lin("features.node_embedding", h, len(POLYTYPES), bias=False)
    ln("features.norm_nodes")
    lin("features.edge_embedding", h, EDGE_IN, bias=False)
    ln("features.norm_edges")
    lin("W_e", h, h)
    s["W_s.weight"] = (vocab, h)
    for i in range(num_encoder_layers):
        p = f"encoder_layers.{i}."
        ln(p + "norm1"); ln(p + "norm2"); ln(p + "norm3")
        lin(p + "W1", h, 3 * h); lin(p + "W2", h, h); lin(p + "W3", h, h)
        lin(p + "W11", h, 3 * h); lin(p + "W12", h, h); lin(p + "W13", h, h)
        lin(p + "dense.W_in", 4 * h, h); lin(p + "dense.W_out", h, 4 * h)
    for i in range(num_decoder_layers):
        p = f"decoder_layers.{i}."
        ln(p + "norm1"); ln(p + "norm2")
        lin(p + "W1", h, 4 * h); lin(p + "W2", h, h); lin(p + "W3", h, h)
        lin(p + "dense.W_in", 4 * h, h); lin(p + "dense.W_out", h, 4 * h)
    lin("W_out", num_letters, h)
    return s
