"""The reference's neighbour-indexing helpers (inference/model_utils.py:707-732 == na_model_utils.py:168-193) with the
same names and argument meaning, on the HIP gather kernels (`namp_gather_*`): pure copies, bit-exact.

    gather_edges(edges [B,N,N,C], neighbor_idx [B,N,K])              -> [B,N,K,C]
    gather_nodes(nodes [B,N,C], neighbor_idx [B,N,K])                -> [B,N,K,C]
    gather_nodes_t(nodes [B,N,C], neighbor_idx [B,K])                -> [B,K,C]
    cat_neighbors_nodes(h_nodes [B,N,C2], h_neighbors [B,N,K,C1], E_idx [B,N,K]) -> [B,N,K,C1+C2]

Any 4-byte element type is accepted (the mask / offset gathers of the reference run on integer tensors: a gather only
moves bits); other dtypes are converted to float32 first.  Tensors must live on the HIP device (no CPU fallback).
"""
from __future__ import annotations

import torch

from . import hip


def _bits(t):
    """-> (contiguous float32 view of the same bits, dtype to view the result back as or None)."""
    if not t.is_cuda:
        raise RuntimeError("na_mpnn_amd.ops: tensors must be on a HIP device (no CPU fallback)")
    if t.dtype == torch.float32:
        return t.contiguous(), None
    if t.element_size() == 4:
        return t.contiguous().view(torch.float32), t.dtype
    return t.float().contiguous(), None


def _back(out, dt):
    return out if dt is None else out.view(dt)


def gather_nodes(nodes, neighbor_idx):
    B, N, K = neighbor_idx.shape
    src, dt = _bits(nodes)
    C = src.shape[2]
    idx = neighbor_idx.to(torch.int32).contiguous()
    out = torch.empty(B, N, K, C, dtype=torch.float32, device=src.device)
    hip.check(hip.lib().namp_gather_nodes_f32(src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, N, K, C, hip.current_stream()),
              "gather_nodes")
    return _back(out, dt)


def gather_edges(edges, neighbor_idx):
    B, N, K = neighbor_idx.shape
    src, dt = _bits(edges)
    C = src.shape[3]
    idx = neighbor_idx.to(torch.int32).contiguous()
    out = torch.empty(B, N, K, C, dtype=torch.float32, device=src.device)
    hip.check(hip.lib().namp_gather_edges_f32(src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, N, K, C, hip.current_stream()),
              "gather_edges")
    return _back(out, dt)


def gather_nodes_t(nodes, neighbor_idx):
    """One index list per batch: nodes [B,N,C], neighbor_idx [B,K] -> [B,K,C]."""
    B, K = neighbor_idx.shape
    src, dt = _bits(nodes)
    N, C = src.shape[1], src.shape[2]
    idx = neighbor_idx.to(torch.int32).contiguous()
    out = torch.empty(B, K, C, dtype=torch.float32, device=src.device)
    hip.check(hip.lib().namp_gather_rows_f32(src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, N, K, C, hip.current_stream()),
              "gather_nodes_t")
    return _back(out, dt)


def cat_neighbors_nodes(h_nodes, h_neighbors, E_idx):
    B, N, K = E_idx.shape
    nodes, dt = _bits(h_nodes)
    nbrs, dt2 = _bits(h_neighbors)
    if dt != dt2:
        nodes, nbrs, dt = h_nodes.float().contiguous(), h_neighbors.float().contiguous(), None
    C1, C2 = nbrs.shape[3], nodes.shape[2]
    idx = E_idx.to(torch.int32).contiguous()
    out = torch.empty(B, N, K, C1 + C2, dtype=torch.float32, device=nodes.device)
    hip.check(hip.lib().namp_cat_neighbors_nodes_f32(nodes.data_ptr(), nbrs.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                                     B, N, K, C1, C2, hip.current_stream()), "cat_neighbors_nodes")
    return _back(out, dt)
