"""ctypes binding of libnamp_hip.so (include/namp.h).

The library is built in-tree by ``na_mpnn_amd.build`` (hipcc, gfx950).  There is no
fallback: if the shared object is missing or a symbol is absent, importing callers get
a RuntimeError that says how to build it.  Raw device pointers (``tensor.data_ptr()``)
and the current HIP stream handle cross the boundary; no torch types do.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NAMP_LIB_PATH") or os.path.join(_HERE, "lib", "libnamp_hip.so")   # env: tools/kbench.py ablations

NAMP_ABI_VERSION = 4
NAMP_MAX_LAYERS = 8
NAMP_FLAG_BF16 = 1
NAMP_FLAG_X3 = 2

c_fp = C.c_void_p   # const float*  (device)
c_ip = C.c_void_p   # const int32_t* (device)


def _fields(names):
    return [(n, c_fp) for n in names]


class NampEncLayerW(C.Structure):
    _fields_ = _fields(["W1a_img", "W1b_img", "W1c_img", "b1", "W2_img", "b2", "W3_img", "b3",
                        "W11a_img", "W11b_img", "W11c_img", "b11", "W12_img", "b12", "W13_img", "b13",
                        "Win_img", "b_in", "Wout_img", "b_out",
                        "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g", "ln3_b",
                        "W1b_bimg", "W2_bimg", "W3_bimg", "W11b_bimg", "W12_bimg", "W13_bimg",
                        "W1b_ximg", "W2_ximg", "W3_ximg", "W11b_ximg", "W12_ximg", "W13_ximg",
                        "Win_ximg", "Wout_ximg", "W1a_ximg", "W1c_ximg", "W11a_ximg", "W11c_ximg",
                        "W1b_simg", "W2_simg", "W11b_simg", "W12_simg", "W13_simg"]) + [("flags", C.c_int64)]


class NampDecLayerW(C.Structure):
    _fields_ = _fields(["W1a_img", "W1e_img", "W1s_img", "W1v_img", "b1", "tok",
                        "W2_img", "b2", "W3_img", "b3", "Win_img", "b_in", "Wout_img", "b_out",
                        "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                        "W1e_bimg", "W2_bimg", "W3_bimg", "W1e_ximg", "W2_ximg", "W3_ximg",
                        "Win_ximg", "Wout_ximg", "W1a_ximg", "W1v_ximg", "W1e_simg", "W2_simg"]) + [("flags", C.c_int64)]


class NampFeatW(C.Structure):
    _fields_ = _fields(["Wedge_img", "pos_w", "pos_b", "ln_g", "ln_b", "Wedge_ximg"])


class NampModelW(C.Structure):
    _fields_ = [("n_enc", C.c_int32), ("n_dec", C.c_int32), ("vocab", C.c_int32), ("reserved", C.c_int32),
                ("Wv_img", c_fp), ("Wv_b", c_fp), ("We_img", c_fp), ("We_b", c_fp),
                ("Wout_w", c_fp), ("Wout_b", c_fp),
                ("enc", NampEncLayerW * NAMP_MAX_LAYERS), ("dec", NampDecLayerW * NAMP_MAX_LAYERS),
                ("feat", NampFeatW), ("We_ximg", c_fp), ("Wv_ximg", c_fp), ("We_bimg", c_fp), ("We_simg", c_fp)]


class NampProj(C.Structure):
    _fields_ = [("img", c_fp), ("bias", c_fp), ("tok", c_fp), ("out", c_fp)]


class NampPack(C.Structure):
    _fields_ = [("W", c_fp), ("img", c_fp), ("ld", C.c_int), ("out_f", C.c_int), ("in_f", C.c_int), ("kind", C.c_int),
                ("transposed", C.c_int), ("first_block", C.c_int)]


class NampReduce(C.Structure):
    _fields_ = [("src", c_fp), ("dst", c_fp), ("A", C.c_longlong), ("Mb", C.c_longlong), ("sa", C.c_longlong), ("sn", C.c_longlong),
                ("n", C.c_int), ("reserved", C.c_int)]


i32, vp, sz = C.c_int, C.c_void_p, C.c_size_t
_PROTOTYPES = {
    # name: (restype, argtypes)      — must list every symbol include/namp.h declares
    "namp_abi_version": (i32, []),
    "namp_last_error": (C.c_char_p, []),
    "namp_pack_image": (i32, [c_fp, i32, i32, i32, i32, c_fp, vp]),
    "namp_pack_image_bf16": (i32, [c_fp, i32, i32, c_fp, vp]),
    "namp_pack_image_bf16_32": (i32, [c_fp, i32, i32, c_fp, vp]),
    "namp_bf16s_message": (i32, [i32, vp, c_ip, c_ip, c_ip, vp, vp, vp, vp, vp, c_fp, c_fp, i32, i32, i32, i32, vp]),
    "namp_pack_image_x3": (i32, [c_fp, i32, i32, c_fp, vp]),
    "namp_pack_feat_x3": (i32, [c_fp, i32, c_fp, vp]),
    "namp_pack_image_x3_general": (i32, [c_fp, i32, i32, i32, i32, c_fp, vp]),
    "namp_gather_nodes_f32": (i32, [c_fp, c_ip, c_fp, i32, i32, i32, i32, vp]),
    "namp_gather_rows_f32": (i32, [c_fp, c_ip, c_fp, C.c_long, i32, i32, i32, vp]),
    "namp_gather_edges_f32": (i32, [c_fp, c_ip, c_fp, i32, i32, i32, i32, vp]),
    "namp_cat_neighbors_nodes_f32": (i32, [c_fp, c_fp, c_ip, c_fp, i32, i32, i32, i32, i32, vp]),
    "namp_node_linear": (i32, [c_fp, c_ip, i32, i32, i32, C.POINTER(NampProj), i32, C.POINTER(NampProj), vp]),
    "namp_edge_embed": (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, vp]),
    "namp_edge_embed_prec": (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, vp]),
    "namp_node_linear_prec": (i32, [c_fp, i32, C.POINTER(NampProj), i32, i32, vp]),
    "namp_enc_message": (i32, [C.POINTER(NampEncLayerW), c_fp, c_ip, c_ip, c_ip, c_fp, c_fp, c_fp, i32, i32, i32, vp]),
    "namp_enc_edge_update": (i32, [C.POINTER(NampEncLayerW), c_fp, c_ip, c_fp, c_fp, c_fp, i32, i32, i32, vp]),
    "namp_node_update": (i32, [c_fp] * 8 + [c_fp, c_fp, c_fp, c_fp, c_ip, c_fp, C.POINTER(NampProj), i32, c_ip, i32, i32, vp]),
    "namp_dec_message": (i32, [C.POINTER(NampDecLayerW), c_fp, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp,
                               i32, i32, i32, i32, vp]),
    "namp_enc_message_update": (i32, [C.POINTER(NampEncLayerW), c_fp, c_ip, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp,
                                      C.POINTER(NampProj), i32, i32, i32, i32, vp]),
    "namp_enc_edge_message_update": (i32, [C.POINTER(NampEncLayerW), c_fp, c_fp, c_fp, C.POINTER(NampEncLayerW), c_ip, c_ip, c_ip,
                                           c_fp, c_fp, c_fp, c_fp, C.POINTER(NampProj), i32, i32, i32, i32, vp]),
    "namp_dec_message_update": (i32, [C.POINTER(NampDecLayerW), c_fp, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp, c_ip, c_fp,
                                      C.POINTER(NampProj), i32, c_ip, c_fp, c_fp, c_fp, c_fp, i32,
                                      i32, i32, i32, i32, vp]),
    "namp_fused_tail_max_residues": (i32, []),
    "namp_logits_log_softmax": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, vp]),
    "namp_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "namp_sample_work_lists": (i32, [c_ip, c_ip, c_ip, c_ip, i32, i32, vp]),
    "namp_decoding_order": (i32, [c_fp, c_fp, c_fp, vp, c_ip, c_ip, i32, i32, i32, vp]),
    "namp_sample_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "namp_sample_workspace_bytes_n": (sz, [i32, i32, i32, i32, i32]),
    "namp_featurize_workspace_bytes": (sz, [i32, i32]),
    "namp_featurize_split_bytes": (sz, [i32, i32, i32]),
    "namp_featurize": (i32, [C.POINTER(NampModelW), c_fp, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, i32, i32, c_ip, c_fp, c_fp,
                             vp, sz, i32, i32, vp]),
    "namp_featurize_ordered": (i32, [C.POINTER(NampModelW), c_fp, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip, i32, i32, c_ip, c_fp, c_fp,
                                     vp, sz, i32, i32, c_fp, c_fp, c_fp, vp, c_ip, c_ip, i32, vp]),
    "namp_decoder_sample": (i32, [C.POINTER(NampModelW), c_fp, c_fp, c_ip, c_ip, c_ip, c_ip, c_ip, c_fp, c_ip, c_ip, c_fp, c_ip,
                                  c_ip, c_ip, c_fp, c_fp,
                                  C.c_float, C.c_uint64, c_ip, c_fp, c_fp, vp, sz, i32, i32, i32, i32, vp]),
    "namp_train_edge_fwd": (i32, [i32, c_fp, c_ip, c_ip, c_ip, c_ip] + [c_fp] * 10 + [C.c_float, C.c_uint32, c_fp, i32, i32, i32, i32, vp]),
    "namp_train_edge_update_bwd_groups": (i32, [i32, i32, i32]),
    "namp_train_edge_update_bwd": (i32, [c_fp, c_ip] + [c_fp] * 11 + [C.c_float, C.c_uint32, C.c_long] + [c_fp] * 10 + [i32, i32, i32, i32, vp]),
    "namp_train_edge_update_bwd_dw": (i32, [c_fp, c_ip] + [c_fp] * 11 + [C.c_float, C.c_uint32] + [c_fp] * 8 + [i32, i32, i32, i32, vp]),
    "namp_train_edge_bwd": (i32, [i32, c_fp, c_ip, c_ip, c_ip, c_ip] + [c_fp] * 22 + [i32, i32, i32, i32, vp]),
    "namp_train_edge_bwd_dw_groups": (i32, [i32, i32, i32]),
    "namp_train_edge_bwd_dw_rows": (C.c_long, [i32, i32, i32]),
    "namp_train_edge_bwd_dw": (i32, [i32, c_fp, c_ip, c_ip, c_ip, c_ip] + [c_fp] * 15 + [i32, i32, i32, i32, vp]),
    "namp_train_scatter_rows": (i32, [c_fp, c_ip, c_ip, vp, c_fp, c_fp, i32, vp]),
    "namp_train_scatter_rows_bf16": (i32, [c_fp, c_ip, c_ip, vp, c_fp, c_fp, i32, vp]),
    "namp_train_tail_groups": (i32, [i32]),
    "namp_train_tail_fwd": (i32, [c_fp, c_fp, c_ip] + [c_fp] * 8 + [C.c_float, C.c_uint32, C.c_uint32] + [c_fp] * 4 + [i32, vp]),
    "namp_train_tail_bwd": (i32, [c_fp, c_fp, c_ip] + [c_fp] * 4 + [C.c_float, C.c_uint32, C.c_uint32] + [c_fp] * 10 + [i32, vp]),
    "namp_reduce_sum": (i32, [C.POINTER(NampReduce), i32, vp]),
    "namp_node_linear_sum": (i32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), i32, c_fp, i32, i32, vp]),
    "namp_edge_embed_ln": (i32, [c_fp] * 6 + [i32, i32, i32, i32, vp]),
    "namp_train_embed_ln_bwd_groups": (i32, [C.c_long]),
    "namp_train_embed_ln_bwd": (i32, [c_fp] * 7 + [vp, i32, C.c_long, vp]),
    "namp_train_wgrad_ln": (i32, [c_fp] * 5 + [i32, C.c_long, c_fp, c_fp, vp]),
    "namp_train_rows_groups": (i32, [C.c_long]),
    "namp_train_class_sums": (i32, [c_fp, c_ip, i32, C.c_long, c_fp, vp]),
    "namp_train_wcolsum": (i32, [c_fp, c_fp, C.c_long, c_fp, vp]),
    "namp_train_reverse_adjacency": (i32, [c_ip, c_ip, c_ip, c_ip, i32, i32, i32, vp]),
    "namp_train_pos_features": (i32, [c_ip, c_ip, c_ip, c_fp, c_fp, c_ip, c_fp, i32, i32, i32, vp]),
    "namp_train_pos_grad_groups": (i32, [C.c_long]),
    "namp_train_pos_grad": (i32, [c_fp, c_fp, i32, c_ip, c_fp, C.c_long, vp]),
    "namp_pack_images": (i32, [c_fp, i32, i32, vp]),
    "namp_train_ln_rows_groups": (i32, [C.c_long]),
    "namp_train_ln_rows_fwd": (i32, [c_fp, c_fp, c_fp, c_fp, C.c_long, vp]),
    "namp_train_ln_rows_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_long, vp]),
    "namp_train_wgrad_chunks": (i32, [C.c_long]),
    "namp_train_wgrad": (i32, [c_fp, c_fp, i32, i32, C.c_long, c_fp, c_fp, vp]),
    "namp_train_wgrad_multi": (i32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), i32, i32, C.c_long, i32, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), vp]),
    "namp_train_feat_wgrad_chunks": (i32, [C.c_long]),
    "namp_train_feat_wgrad_ws_ints": (C.c_long, [C.c_long]),
    "namp_train_feat_wgrad": (i32, [c_fp, c_fp, c_ip, c_fp, c_fp, vp, c_fp, c_ip, i32, i32, i32, i32, vp]),
    "namp_train_g16_elems": (C.c_long, [C.c_long]),
    "namp_train_loss_smoothed": (i32, [i32, c_ip, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.POINTER(C.c_float), C.c_double, c_ip, vp,
                                       vp, vp, c_fp, C.c_long, i32, vp]),
    "namp_train_adam_chunk": (i32, []),
    "namp_train_adam_step": (i32, [vp, vp, vp, vp, i32, i32, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_float, c_fp, vp]),
    "namp_sample_levels": (i32, [c_ip, c_ip, c_ip, c_ip, i32, i32, i32, i32, vp]),
    "namp_sample_levels_dep": (i32, [c_ip, c_ip, c_ip, c_ip, i32, c_ip, c_ip, c_ip, i32, i32, i32, i32, vp]),
    "namp_decoder_sample_levels": (i32, [C.POINTER(NampModelW), c_fp, c_fp, c_ip, c_ip, c_ip, c_ip, c_ip, c_fp, c_ip, c_ip, c_fp, c_ip,
                                         c_ip, c_ip, c_fp, c_fp,
                                         c_ip, c_ip, C.POINTER(C.c_int32), i32,
                                         C.c_float, C.c_uint64, c_ip, c_fp, c_fp, vp, sz, i32, i32, i32, i32, vp]),
    "namp_decoder_sample_walk_grid": (i32, [i32, i32, i32]),
    "namp_decoder_sample_walk": (i32, [C.POINTER(NampModelW), c_fp, c_fp, c_ip, c_ip, c_ip, c_ip, c_ip, c_fp, c_ip, c_ip, c_fp, c_ip,
                                       c_ip, c_ip, c_fp, c_fp,
                                       c_ip, c_ip, i32, c_ip, c_ip, c_ip, c_fp,
                                       C.c_float, C.c_uint64, c_ip, c_fp, c_fp, vp, sz, i32, i32, i32, i32, vp]),
    "namp_profile_enable": (i32, [i32]),
    "namp_profile_collect": (i32, [C.POINTER(C.c_float), C.POINTER(C.c_int32), i32]),
    "namp_enc_layer_fwd": (i32, [C.POINTER(NampEncLayerW), c_fp, c_fp, c_ip, c_ip, c_ip, c_fp, c_fp,
                                 vp, sz, i32, i32, i32, vp]),
    "namp_set_persistent": (i32, [i32]),
    "namp_set_bf16p": (i32, [i32]),
    "namp_persistent_status": (i32, [vp, sz, i32, i32, i32, C.POINTER(C.c_int32)]),
    "namp_dec_layer_fwd": (i32, [C.POINTER(NampDecLayerW), c_fp, c_fp, c_ip, c_fp, c_fp, vp, sz, i32, i32, i32, vp]),
    "namp_encoder_fwd": (i32, [C.POINTER(NampModelW), c_fp, c_fp, c_ip, c_ip, c_fp, c_fp,
                               vp, sz, i32, i32, i32, vp]),
    "namp_encdec_fwd": (i32, [C.POINTER(NampModelW), c_fp, c_fp, c_ip, c_ip, c_ip, c_ip, c_fp, c_fp, c_fp, c_fp,
                              vp, sz, i32, i32, i32, vp]),
    "namp_decoder_fwd": (i32, [C.POINTER(NampModelW), c_fp, c_fp, c_ip, c_ip, c_ip, c_ip, c_fp, c_fp, c_fp,
                               vp, sz, i32, i32, i32, i32, vp]),
}

KERNEL_KINDS = ["gather", "node_linear", "edge_embed", "enc_message", "enc_edge_update", "node_update",
                "dec_message", "logits", "features", "enc_edge_message", "enc_edge_dec_message", "encdec_persistent"]

_lib = None


def exported_symbols():
    return sorted(_PROTOTYPES)


def lib():
    """The loaded library with prototypes set.  Raises (never falls back) if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must bring in ITS HIP runtime first: the process may hold only one libamdhip64, and the device
    # pointers / streams handed to this library come from torch's.  (Loading this .so first binds /opt/rocm's
    # runtime instead and every call then fails with "no ROCm-capable device is detected".)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -m na_mpnn_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _PROTOTYPES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild with `python -m na_mpnn_amd.build`") from e
        fn.restype, fn.argtypes = res, args
    v = L.namp_abi_version()
    if v != NAMP_ABI_VERSION:
        raise RuntimeError(f"libnamp_hip.so ABI version {v} != expected {NAMP_ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().namp_last_error().decode(errors="replace")
        raise RuntimeError(f"libnamp_hip {what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def profile_collect():
    """{kind: (total_ms, launches)} recorded since namp_profile_enable(1)."""
    n = len(KERNEL_KINDS)
    ms, cnt = (C.c_float * n)(), (C.c_int32 * n)()
    check(lib().namp_profile_collect(ms, cnt, n), "profile_collect")
    return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNEL_KINDS)}
