"""Multi-process (world_size 2, gloo, CPU) test of the N>1 path: LPT sharding of independent
complexes + the reporting all-gather (na_mpnn_amd/shard.py).  No data-path collective exists."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from na_mpnn_amd import shard


def fake_result(i, n):
    return (torch.arange(n, dtype=torch.int64) * (i + 3)) % 33


def _worker(rank, world, port, lengths, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = shard.lpt_assign(lengths, world)
    local = {i: fake_result(i, int(lengths[i])) for i in shards[rank]}
    out = shard.all_gather_ragged(local, len(lengths))
    ok = all(o is not None and torch.equal(o.long(), fake_result(i, int(lengths[i]))) for i, o in enumerate(out))
    q.put((rank, ok, len(local)))
    dist.barrier()
    dist.destroy_process_group()


def test_lpt_balance_and_coverage():
    lengths = shard.synthetic_lengths(1373)
    assert len(lengths) == 1373 and lengths.min() >= 50 and lengths.max() <= 6000
    shards = shard.lpt_assign(lengths, 8)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(1373))                       # every complex exactly once
    loads = np.array([lengths[s].sum() for s in shards])
    assert loads.max() / loads.mean() < 1.01               # LPT keeps the 8 GPUs within 1 %
    assert shard.lpt_assign([5, 1, 1], 1) == [[0, 1, 2]]
    assert shard.lpt_assign([], 2) == [[], []]             # empty input


def test_single_process_collation():
    local = {2: fake_result(2, 7), 0: fake_result(0, 3)}
    out = shard.all_gather_ragged(local, 4)
    assert out[1] is None and out[3] is None
    assert torch.equal(out[0].long(), fake_result(0, 3)) and torch.equal(out[2].long(), fake_result(2, 7))


def test_two_rank_gloo_all_gather():
    lengths = np.array([40, 7, 300, 12, 95, 1, 64, 200, 33])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == len(lengths)


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.LayerNorm(7), torch.nn.Linear(7, 3))
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1)) if i != 2 else None     # one parameter without a gradient
    n = shard.allreduce_gradients(net.parameters())
    expect = [(sum(r + 1 for r in range(world)) / world) * (i + 1) if i != 2 else 0.0 for i in range(len(list(net.parameters())))]
    ok = all(torch.allclose(p.grad, torch.full_like(p, e)) for p, e in zip(net.parameters(), expect))
    q.put((rank, ok, n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce():
    """The data-parallel training exchange: one bucketed all-reduce averages every parameter's gradient over the ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and all(n == 5 * 7 + 7 + 7 + 7 + 7 * 3 + 3 for _, _, n in res)
    assert shard.allreduce_gradients(torch.nn.Linear(2, 2).parameters()) == 0          # no process group: nothing to do

