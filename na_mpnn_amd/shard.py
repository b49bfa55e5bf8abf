"""Sharding of independent complexes over the GPUs of a node.

The reference scales out by running independent structures in separate SLURM array tasks
(/root/reference/evaluation/rna_design_scripts/design_sequences.sh:41-50); there is no exchange
inside the path.  Here: one process per GPU, complexes assigned by longest-processing-time-first
on their residue counts, NO collective on the data path, and one all-gather at the end that collates
the designed sequences (or any per-residue int tensor) on every rank for reporting.  The collective
goes through ``torch.distributed`` — backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import numpy as np
import torch


def lpt_assign(lengths, world_size: int):
    """Longest-processing-time-first: returns a list (per rank) of complex indices.
    Cost model: residues (the path is O(N*K) with K fixed)."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += int(lengths[i])
    return shards


def synthetic_lengths(n_complexes: int = 1373, seed: int = 4, lo: int = 50, hi: int = 3000, cap: int = 6000):
    """cfg4 size distribution: N_i = round(exp(U(ln lo, ln hi))), capped.
    1373 = len(splits/design_test.json) — the split holds PDB ids only, no coordinates."""
    rng = np.random.default_rng(seed)
    n = np.round(np.exp(rng.uniform(np.log(lo), np.log(hi), n_complexes))).astype(np.int64)
    return np.minimum(n, cap)


def token_batches(lengths, indices=None, max_tokens: int = 6000, keep_oversize: bool = True):
    """Length-sorted token-bucket batches, the rule of the reference's StructureLoader (na_data_utils.py:1405-1426):
    walk the complexes by ascending length and close a batch when ``length * (batch size + 1)`` would exceed
    ``max_tokens``.  The reference drops complexes longer than the budget (training); inference keeps them as
    singleton batches (``keep_oversize``).  Returns a list of index lists."""
    idx = np.arange(len(lengths)) if indices is None else np.asarray(indices, dtype=np.int64)
    order = idx[np.argsort(np.asarray(lengths)[idx], kind="stable")]
    out, batch = [], []
    for ix in order:
        size = int(lengths[ix])
        if size > max_tokens:
            if keep_oversize:
                out.append([int(ix)])
            continue
        if size * (len(batch) + 1) <= max_tokens:
            batch.append(int(ix))
        else:
            if batch:
                out.append(batch)
            batch = [int(ix)]
    if batch:
        out.append(batch)
    return out


# padding values of the reference's batch featurize (na_model_utils.py:14-35)
PAD_VALUES = {"S": 32, "R_polymer_type": 5, "R_idx": -100, "chain_labels": -1}


def pad_batch(complexes, device=None):
    """Stack per-complex numpy dicts (no batch dimension) into one padded feature_dict: zeros everywhere (mask = 0 on
    the padding) except S = PAD, R_polymer_type = PAD, R_idx = -100, chain_labels = -1."""
    L = max(c["X"].shape[0] for c in complexes)
    fd = {}
    for key in complexes[0]:
        rows = []
        for c in complexes:
            a = np.asarray(c[key])
            p = np.full((L - a.shape[0],) + a.shape[1:], PAD_VALUES.get(key, 0), dtype=a.dtype)
            rows.append(np.concatenate([a, p], 0))
        t = torch.from_numpy(np.stack(rows))
        fd[key] = t.to(device) if device is not None else t
    return fd


def all_gather_ragged(local: dict, n_total: int, device=None, group=None):
    """Collate {complex index: 1-D int tensor} from every rank.

    Two collectives: lengths (int64 [n_total], summed: each index is owned by exactly one rank) and one
    padded all-gather of the concatenated payload.  Returns a list of n_total tensors (None where no
    rank produced a result)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    device = device or (next(iter(local.values())).device if local else torch.device("cpu"))
    lens = torch.zeros(n_total, dtype=torch.int64, device=device)
    owner = torch.full((n_total,), -1, dtype=torch.int64, device=device)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    keys = sorted(local)
    for k in keys:
        lens[k] = local[k].numel()
        owner[k] = rank
    payload = torch.cat([local[k].reshape(-1).to(device=device, dtype=torch.int32) for k in keys]) if keys else \
        torch.zeros(0, dtype=torch.int32, device=device)
    if world == 1:
        out, off = [None] * n_total, 0
        for k in keys:
            out[k] = payload[off:off + int(lens[k])]
            off += int(lens[k])
        return out
    dist.all_reduce(lens, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(owner, op=dist.ReduceOp.MAX, group=group)
    per_rank = torch.zeros(world, dtype=torch.int64, device=device)
    per_rank[rank] = payload.numel()
    dist.all_reduce(per_rank, op=dist.ReduceOp.SUM, group=group)
    width = int(per_rank.max())
    padded = torch.zeros(width, dtype=torch.int32, device=device)
    padded[:payload.numel()] = payload
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    out = [None] * n_total
    offs = [0] * world
    owner_l, lens_l = owner.tolist(), lens.tolist()
    for k in range(n_total):            # every rank packed its keys in ascending order
        r = owner_l[k]
        if r < 0:
            continue
        out[k] = gathered[r][offs[r]:offs[r] + lens_l[k]]
        offs[r] += lens_l[k]
    return out


def allreduce_gradients(parameters, group=None, average: bool = True):
    """Data-parallel training step's exchange (an extension: the reference trains on one GPU, na_run.py): all ranks hold
    the same weights and their own batch; the gradients of all parameters are packed into ONE contiguous fp32 bucket
    (2.3 M values = 9.2 MB — far below what a ring over the 7 x 153 GB/s xGMI links needs to be bandwidth-bound, so a
    single all-reduce beats any bucketing), summed with one RCCL all-reduce, averaged and scattered back.
    Parameters without a gradient contribute zeros (every rank must call with the same parameter list)."""
    import torch.distributed as dist
    params = [p for p in parameters]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    dev = params[0].device
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
    return flat.numel()

