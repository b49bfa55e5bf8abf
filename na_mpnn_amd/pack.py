"""Weight pre-packing: reference ``state_dict`` -> device buffers in the layout the kernels read.

Host logic only (offset planning) plus calls into libnamp_hip.so that run on the device:
every [128 x 128] block of an ``nn.Linear`` weight becomes a 64 KiB MFMA-fragment image
(``namp_pack_image``), vectors are copied verbatim, and the decoder's per-token table
``W_s.weight @ W1[:, 256:384].T`` is produced by ``namp_node_linear``.

Block split of the concatenated first-layer inputs (reference channel orders):
  EncLayer  W1 / W11 [128, 384] = [ h_V_i | h_E_ik | h_V_j ]            (model_utils.py:684-686)
  DecLayer  W1       [128, 512] = [ h_V_i | h_E_ik | h_S_j | h_V_j ]    (model_utils.py:407,416,640-641)
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

from . import hip, spec

H = spec.H
ALIGN = 64  # floats (256 B)


def _round(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


def plan(n_enc: int, n_dec: int, vocab: int):
    """Ordered {item: (offset_floats, n_floats, recipe)} of the flat packed buffer.

    recipe = ("img", key, col0, out_f, in_f) | ("vec", key) | ("tok", layer)
    """
    items = OrderedDict()
    off = 0

    def add(name, n, recipe):
        nonlocal off
        items[name] = (off, n, recipe)
        off += _round(n)

    def img(name, key, col0, out_f=H, in_f=H):
        add(name, out_f * in_f, ("img", key, col0, out_f, in_f))

    def vec(name, key, n):
        add(name, n, ("vec", key))

    def bimg(name, key, col0):                      # 32 KiB bf16 image = 8192 float slots
        add(name, H * H // 2, ("bimg", key, col0))

    def simg(name, key, col0):                      # 32 KiB bf16 image in the 32x32x16 operand order
        add(name, H * H // 2, ("simg", key, col0))

    def ximg(name, key, col0):                      # 64 KiB x3 image (bf16 hi | bf16 mid) = 16384 float slots
        add(name, H * H, ("ximg", key, col0))

    def xgimg(name, key, out_f, in_f):              # x3 image of a general block (residue-level FFN weights)
        add(name, out_f * in_f, ("xgimg", key, out_f, in_f))

    img("Wv_img", "W_v.weight", 0); vec("Wv_b", "W_v.bias", H)
    ximg("We_ximg", "W_e.weight", 0); ximg("Wv_ximg", "W_v.weight", 0); bimg("We_bimg", "W_e.weight", 0)
    simg("We_simg", "W_e.weight", 0)
    img("We_img", "W_e.weight", 0); vec("We_b", "W_e.bias", H)
    vec("Wout_w", "W_out.weight", vocab * H); vec("Wout_b", "W_out.bias", vocab)
    # featuriser (ProteinFeaturesNA): 5200-wide edge embedding as a 325-k-tile image
    img("feat.Wedge_img", "features.edge_embedding.weight", 0, H, spec.EDGE_IN)
    add("feat.Wedge_ximg", H * spec.EDGE_IN, ("fximg", "features.edge_embedding.weight"))
    vec("feat.pos_w", "features.embeddings.linear.weight", spec.NUM_POS * (2 * spec.MAX_REL + 2))
    vec("feat.pos_b", "features.embeddings.linear.bias", spec.NUM_POS)
    vec("feat.ln_g", "features.norm_edges.weight", H); vec("feat.ln_b", "features.norm_edges.bias", H)
    for l in range(n_enc):
        p, q = f"enc{l}.", f"encoder_layers.{l}."
        for nm, c0 in (("W1a", 0), ("W1b", H), ("W1c", 2 * H)):
            img(p + nm + "_img", q + "W1.weight", c0)
        vec(p + "b1", q + "W1.bias", H)
        img(p + "W2_img", q + "W2.weight", 0); vec(p + "b2", q + "W2.bias", H)
        img(p + "W3_img", q + "W3.weight", 0); vec(p + "b3", q + "W3.bias", H)
        for nm, c0 in (("W11a", 0), ("W11b", H), ("W11c", 2 * H)):
            img(p + nm + "_img", q + "W11.weight", c0)
        vec(p + "b11", q + "W11.bias", H)
        img(p + "W12_img", q + "W12.weight", 0); vec(p + "b12", q + "W12.bias", H)
        img(p + "W13_img", q + "W13.weight", 0); vec(p + "b13", q + "W13.bias", H)
        img(p + "Win_img", q + "dense.W_in.weight", 0, 4 * H, H); vec(p + "b_in", q + "dense.W_in.bias", 4 * H)
        img(p + "Wout_img", q + "dense.W_out.weight", 0, H, 4 * H); vec(p + "b_out", q + "dense.W_out.bias", H)
        for i in (1, 2, 3):
            vec(p + f"ln{i}_g", q + f"norm{i}.weight", H); vec(p + f"ln{i}_b", q + f"norm{i}.bias", H)
        bimg(p + "W1b_bimg", q + "W1.weight", H); bimg(p + "W2_bimg", q + "W2.weight", 0); bimg(p + "W3_bimg", q + "W3.weight", 0)
        bimg(p + "W11b_bimg", q + "W11.weight", H); bimg(p + "W12_bimg", q + "W12.weight", 0); bimg(p + "W13_bimg", q + "W13.weight", 0)
        ximg(p + "W1b_ximg", q + "W1.weight", H); ximg(p + "W2_ximg", q + "W2.weight", 0); ximg(p + "W3_ximg", q + "W3.weight", 0)
        ximg(p + "W11b_ximg", q + "W11.weight", H); ximg(p + "W12_ximg", q + "W12.weight", 0); ximg(p + "W13_ximg", q + "W13.weight", 0)
        xgimg(p + "Win_ximg", q + "dense.W_in.weight", 4 * H, H); xgimg(p + "Wout_ximg", q + "dense.W_out.weight", H, 4 * H)
        ximg(p + "W1a_ximg", q + "W1.weight", 0); ximg(p + "W1c_ximg", q + "W1.weight", 2 * H)
        ximg(p + "W11a_ximg", q + "W11.weight", 0); ximg(p + "W11c_ximg", q + "W11.weight", 2 * H)
        simg(p + "W1b_simg", q + "W1.weight", H); simg(p + "W2_simg", q + "W2.weight", 0)
        simg(p + "W11b_simg", q + "W11.weight", H); simg(p + "W12_simg", q + "W12.weight", 0); simg(p + "W13_simg", q + "W13.weight", 0)
    for l in range(n_dec):
        p, q = f"dec{l}.", f"decoder_layers.{l}."
        for nm, c0 in (("W1a", 0), ("W1e", H), ("W1s", 2 * H), ("W1v", 3 * H)):
            img(p + nm + "_img", q + "W1.weight", c0)
        vec(p + "b1", q + "W1.bias", H)
        add(p + "tok", vocab * H, ("tok", l))
        img(p + "W2_img", q + "W2.weight", 0); vec(p + "b2", q + "W2.bias", H)
        img(p + "W3_img", q + "W3.weight", 0); vec(p + "b3", q + "W3.bias", H)
        img(p + "Win_img", q + "dense.W_in.weight", 0, 4 * H, H); vec(p + "b_in", q + "dense.W_in.bias", 4 * H)
        img(p + "Wout_img", q + "dense.W_out.weight", 0, H, 4 * H); vec(p + "b_out", q + "dense.W_out.bias", H)
        for i in (1, 2):
            vec(p + f"ln{i}_g", q + f"norm{i}.weight", H); vec(p + f"ln{i}_b", q + f"norm{i}.bias", H)
        bimg(p + "W1e_bimg", q + "W1.weight", H); bimg(p + "W2_bimg", q + "W2.weight", 0); bimg(p + "W3_bimg", q + "W3.weight", 0)
        ximg(p + "W1e_ximg", q + "W1.weight", H); ximg(p + "W2_ximg", q + "W2.weight", 0); ximg(p + "W3_ximg", q + "W3.weight", 0)
        xgimg(p + "Win_ximg", q + "dense.W_in.weight", 4 * H, H); xgimg(p + "Wout_ximg", q + "dense.W_out.weight", H, 4 * H)
        ximg(p + "W1a_ximg", q + "W1.weight", 0); ximg(p + "W1v_ximg", q + "W1.weight", 3 * H)
        simg(p + "W1e_simg", q + "W1.weight", H); simg(p + "W2_simg", q + "W2.weight", 0)
    return items, off


def image_index(out_f: int, in_f: int):
    """Host restatement of the fragment-image permutation (pack_image_kernel):
    returns int arrays (n, k) such that img.flat[e] = W[n[e], k[e]].  Used by the CPU tests."""
    import numpy as np
    e = np.arange(out_f * in_f)
    r, lane, t = e & 3, (e >> 2) & 63, e >> 8
    ntn = out_f // 16
    tn, tk = t % ntn, t // ntn
    return 16 * tn + (lane & 15), 16 * tk + 4 * (lane >> 4) + r


class PackedWeights:
    """Device-resident packed parameters + the NampModelW pointer table handed to the library."""

    def __init__(self, state_dict, n_enc: int, n_dec: int, vocab: int, device):
        import torch
        self.n_enc, self.n_dec, self.vocab = n_enc, n_dec, vocab
        if n_enc > hip.NAMP_MAX_LAYERS or n_dec > hip.NAMP_MAX_LAYERS:
            raise ValueError(f"at most {hip.NAMP_MAX_LAYERS} encoder/decoder layers are supported")
        self.precision = "x3"            # per-edge GEMMs: "x3" split-bf16 (fp32-equivalent, default) | "fp32" exact MFMA | "bf16"
        self.items, total = plan(n_enc, n_dec, vocab)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.repack(state_dict)

    def addr(self, name):
        return self.flat.data_ptr() + 4 * self.items[name][0]

    def view(self, name):
        off, n, _ = self.items[name]
        return self.flat[off:off + n]

    def repack(self, state_dict):
        """(Re)build all packed buffers from float32 device tensors of ``state_dict``."""
        import torch
        L = hip.lib()
        stream = hip.current_stream()
        dev = self.flat.device
        keep = []   # keep temporaries alive until the stream has consumed them
        def src(key):
            t = state_dict[key].detach()
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t
        for name, (off, n, recipe) in self.items.items():
            if recipe[0] == "img":
                _, key, col0, out_f, in_f = recipe
                w = src(key)
                hip.check(L.namp_pack_image(w.data_ptr(), w.shape[1], col0, out_f, in_f, self.addr(name), stream),
                          f"pack_image({name})")
            elif recipe[0] == "vec":
                self.flat[off:off + n].copy_(src(recipe[1]).reshape(-1))
            elif recipe[0] == "ximg":
                w = src(recipe[1])
                hip.check(L.namp_pack_image_x3(w.data_ptr(), w.shape[1], recipe[2], self.addr(name), stream),
                          f"pack_image_x3({name})")
            elif recipe[0] == "xgimg":
                _, key, out_f, in_f = recipe
                w = src(key)
                hip.check(L.namp_pack_image_x3_general(w.data_ptr(), w.shape[1], 0, out_f, in_f, self.addr(name), stream),
                          f"pack_image_x3_general({name})")
            elif recipe[0] == "fximg":
                w = src(recipe[1])
                hip.check(L.namp_pack_feat_x3(w.data_ptr(), w.shape[1], self.addr(name), stream), f"pack_feat_x3({name})")
            elif recipe[0] == "bimg":
                w = src(recipe[1])
                hip.check(L.namp_pack_image_bf16(w.data_ptr(), w.shape[1], recipe[2], self.addr(name), stream),
                          f"pack_image_bf16({name})")
            elif recipe[0] == "simg":
                w = src(recipe[1])
                hip.check(L.namp_pack_image_bf16_32(w.data_ptr(), w.shape[1], recipe[2], self.addr(name), stream),
                          f"pack_image_bf16_32({name})")
        # per-token tables need the packed W1s images: tok_l = W_s.weight @ W1s_l^T  [vocab,128]
        ws = src("W_s.weight")
        for l in range(self.n_dec):
            proj = hip.NampProj(self.addr(f"dec{l}.W1s_img"), None, None, self.addr(f"dec{l}.tok"))
            hip.check(L.namp_node_linear(ws.data_ptr(), None, 1, 1, self.vocab, C.byref(proj), 1, None, stream),
                      f"tok table {l}")
        torch.cuda.current_stream().synchronize()
        del keep
        self._build_struct()

    def _build_struct(self):
        m = hip.NampModelW()
        m.n_enc, m.n_dec, m.vocab, m.reserved = self.n_enc, self.n_dec, self.vocab, 0
        for f in ("Wv_img", "Wv_b", "We_img", "We_b", "Wout_w", "Wout_b"):
            setattr(m, f, self.addr(f))
        flags = {"bf16": hip.NAMP_FLAG_BF16, "x3": hip.NAMP_FLAG_X3, "fp32": 0}[getattr(self, "precision", "x3")]
        for l in range(self.n_enc):
            for f, _ in hip.NampEncLayerW._fields_:
                setattr(m.enc[l], f, flags if f == "flags" else self.addr(f"enc{l}.{f}"))
        for l in range(self.n_dec):
            for f, _ in hip.NampDecLayerW._fields_:
                setattr(m.dec[l], f, flags if f == "flags" else self.addr(f"dec{l}.{f}"))
        for f, _ in hip.NampFeatW._fields_:
            setattr(m.feat, f, self.addr(f"feat.{f}"))
        if flags == 0:
            m.feat.Wedge_ximg = None                       # exact fp32 MFMA in the featuriser too
        m.We_ximg = self.addr("We_ximg")
        m.Wv_ximg = self.addr("Wv_ximg")
        m.We_bimg = self.addr("We_bimg")
        m.We_simg = self.addr("We_simg")
        self.struct = m

    def set_precision(self, precision: str):
        """"fp32" (parity mode, default) or "bf16" (per-edge GEMMs in bf16: throughput mode, BASELINE configs[2])."""
        if precision not in ("x3", "fp32", "bf16"):
            raise ValueError("precision must be 'x3' (split-bf16, fp32-equivalent: the default), 'fp32' (exact fp32 MFMA) or 'bf16'")
        self.precision = precision
        self._build_struct()

    def enc_layer(self, l):
        return C.byref(self.struct.enc[l])

    def dec_layer(self, l):
        return C.byref(self.struct.dec[l])

    def model(self):
        return C.byref(self.struct)
