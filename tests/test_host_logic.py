"""CPU-side tests: the C-ABI library loads and exports every symbol include/namp.h declares,
host-side planning logic, the drop-in module's parameter surface, and loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from na_mpnn_amd import hip, spec, synth
from na_mpnn_amd.model import ProteinMPNN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "namp.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(namp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(hip.exported_symbols()), declared ^ set(hip.exported_symbols())
    L = hip.lib()                       # loads on a CPU-only box (no compute calls are made)
    for name in declared:
        assert hasattr(L, name)
    assert L.namp_abi_version() == hip.NAMP_ABI_VERSION
    assert L.namp_workspace_bytes(1, 1, 1000, 48) > 0
    assert L.namp_workspace_bytes(0, 1, 10, 4) == 0


def test_argument_validation_needs_no_gpu():
    L = hip.lib()
    rc = L.namp_edge_embed(None, None, None, None, 1, 10, 4, None)
    assert rc == -1 and b"null pointer" in L.namp_last_error()
    rc = L.namp_pack_image(16, 100, 0, 24, 16, 32, None)          # out_f not a multiple of 16
    assert rc == -1 and b"multiples of 16" in L.namp_last_error()
    with pytest.raises(RuntimeError, match="multiples of 16"):
        hip.check(rc, "pack_image")


def test_struct_layout_matches_header():
    """ctypes mirrors of the weight structs: one pointer per field, in header order."""
    header = open(os.path.join(ROOT, "include", "namp.h")).read()
    for cls in (hip.NampEncLayerW, hip.NampDecLayerW):
        body = re.search(r"typedef struct %s \{(.*?)\}" % cls.__name__, header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"\*\s*([A-Za-z0-9_]+)", body) + re.findall(r"int64_t\s+([A-Za-z0-9_]+)", body)
        assert names == [f for f, _ in cls._fields_]


def test_module_surface_and_state_dict():
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=48, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    sd, sp = m.state_dict(), spec.state_dict_spec()
    assert list(sd) and set(sd) == set(sp)
    assert all(tuple(sd[k].shape) == tuple(sp[k]) for k in sp)
    assert sum(v.numel() for v in sd.values()) == 2_293_457
    w = synth.make_weights(0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    for name in ("encode", "score", "sample", "unconditional_probs", "forward"):
        assert callable(getattr(m, name))
    with pytest.raises(Exception, match="atom_dict"):
        ProteinMPNN(polytype_to_int=spec.polytype_to_int())


def test_cpu_tensors_fail_loudly():
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=8, atom_dict=spec.atom_dict(),
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    cx = synth.make_complex(seed=1, n=20)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in cx.items()}
    fd["batch_size"] = 1
    with pytest.raises(RuntimeError, match="HIP device"):
        m.score(fd)
    with pytest.raises(RuntimeError, match="HIP device"):
        m.encode_graph(torch.zeros(1, 4, 128), torch.zeros(1, 4, 2, 128), torch.zeros(1, 4, 2, dtype=torch.long),
                       torch.ones(1, 4))


def test_non_reference_atom_order_raises():
    """There is one featuriser (the HIP kernels, reference atom order run.py:15-19) and no stock-ops fallback behind it."""
    ad = spec.atom_dict()
    ad["N"], ad["CA"] = ad["CA"], ad["N"]
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=8, atom_dict=ad,
                    restype_to_int=spec.restype_to_int(), polytype_to_int=spec.polytype_to_int())
    cx = synth.make_complex(seed=1, n=20)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in cx.items()}
    fd["batch_size"] = 1
    for call in (m.featurize, m.encode, m.score):
        with pytest.raises(NotImplementedError, match="atom order"):
            call(fd)
    assert not hasattr(m, "featurize_torch")


def test_decoding_rank_helpers():
    order = torch.tensor([[2, 0, 3, 1], [1, 2, 3, 0]])
    rank = ProteinMPNN.ranks_of(order)
    assert rank.tolist() == [[1, 3, 0, 2], [3, 0, 1, 2]]
    cm = torch.tensor([[1, 0, 1, 1]])
    rn = torch.tensor([[0.5, -3.0, -0.1, 2.0]])
    assert ProteinMPNN.decoding_order(cm, rn).tolist() == [[1, 2, 0, 3]]     # fixed position first


def test_order_and_rank_stock_path_and_cast_cache():
    """Host side of round 6's glue on CPU tensors: `order_and_rank` falls back to the stock ops off the device (same order as
    `decoding_order`, rank = its inverse, int32), and `_as` converts a tensor once per OBJECT and version: the same tensor returns the cached
    copy, an in-place edit or a new tensor converts afresh, a tensor already in the kernels' dtype is returned as it is, dead sources leave."""
    import gc
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=8, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                    polytype_to_int=spec.polytype_to_int())
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand(1, 40, generator=g) > 0.2).float()
    cm = (torch.rand(1, 40, generator=g) > 0.5).float()
    randn = torch.randn(3, 40, generator=g)
    order, rank = m.order_and_rank(mask, cm, randn)
    ref = ProteinMPNN.decoding_order(mask * cm, randn)
    assert torch.equal(order, ref) and rank.dtype == torch.int32
    assert torch.equal(rank.long(), ProteinMPNN.ranks_of(ref))
    t = torch.arange(6, dtype=torch.int64)
    a = m._as(t, "i32")
    assert a.dtype == torch.int32 and m._as(t, "i32") is a                      # cached per object
    t.add_(1)
    b = m._as(t, "i32")
    assert b is not a and torch.equal(b.long(), t)                              # version bump: converted afresh
    u = torch.arange(6, dtype=torch.int32)
    assert m._as(u, "i32") is u                                                 # already in the kernels' dtype
    n = len(m._conv)
    del t, a, b
    gc.collect()
    assert len(m._conv) < n                                                     # the weak reference's callback removed the entry


def test_synthetic_generators_are_deterministic():
    a, b = synth.make_graph(seed=3, batch=2, n=50, k=16), synth.make_graph(seed=3, batch=2, n=50, k=16)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert (a["E_idx"][:, :, 0] == np.arange(50)).all()              # self is neighbour 0
    g = synth.make_graph(seed=3, batch=1, n=10, k=48)
    assert g["E_idx"].shape == (1, 10, 10)                          # L < K -> K' = L
    w1, w2 = synth.make_weights(0), synth.make_weights(0)
    assert all(np.array_equal(w1[k], w2[k]) for k in w1)


def test_token_batches_follow_the_reference_rule():
    """StructureLoader (na_data_utils.py:1405-1426): ascending lengths, close when size*(n+1) > budget."""
    from na_mpnn_amd import shard
    lengths = np.array([50, 700, 60, 3000, 55, 7000, 650, 2999, 10])
    b = shard.token_batches(lengths, max_tokens=6000)
    assert sorted(i for x in b for i in x) == list(range(9))                 # inference keeps the oversize complex
    assert [5] in b and all(len(x) == 1 or max(lengths[x]) * len(x) <= 6000 for x in b)
    flat = [i for x in b if x != [5] for i in x]
    assert list(lengths[flat]) == sorted(lengths[flat])                       # ascending walk
    assert shard.token_batches(lengths, max_tokens=6000, keep_oversize=False) == [x for x in b if x != [5]]
    # restating the loop literally
    exp, cur = [], []
    for ix in np.argsort(lengths, kind="stable"):
        if lengths[ix] > 6000:
            continue
        if lengths[ix] * (len(cur) + 1) <= 6000:
            cur.append(int(ix))
        else:
            exp.append(cur); cur = [int(ix)]
    exp.append(cur)
    assert shard.token_batches(lengths, max_tokens=6000, keep_oversize=False) == exp
    sub = shard.token_batches(lengths, indices=[1, 6, 0], max_tokens=1400)
    assert sub == [[0, 6], [1]]                  # 50 | 650*2 <= 1400 | 700*3 > 1400


def test_pad_batch_uses_the_reference_padding_values():
    from na_mpnn_amd import shard, synth
    cxs = [synth.make_complex(seed=i, n=n) for i, n in enumerate((12, 20, 7))]
    fd = shard.pad_batch(cxs)
    assert fd["X"].shape == (3, 20, 16, 3) and fd["mask"].shape == (3, 20)
    assert int(fd["mask"][2, 7:].sum()) == 0 and float(fd["X"][0, 12:].abs().sum()) == 0.0
    assert (fd["S"][0, 12:] == 32).all() and (fd["R_polymer_type"][2, 7:] == 5).all()
    assert (fd["R_idx"][0, 12:] == -100).all() and (fd["chain_labels"][2, 7:] == -1).all()
    assert np.array_equal(fd["X"][1].numpy(), cxs[1]["X"])



def _model(**kw):
    args = dict(num_letters=33, vocab=33, k_neighbors=8, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                polytype_to_int=spec.polytype_to_int())
    args.update(kw)
    return ProteinMPNN(**args)


def test_token_range_and_constructor_checks():
    """Token ids index per-token tables inside the kernels: the host wrappers refuse anything outside [0, vocab) like the
    reference's nn.Embedding does; vocab != num_letters and too many layers are refused at construction."""
    m = _model()
    m._check_tokens(torch.tensor([[0, 5, 32]]))
    with pytest.raises(IndexError, match=r"\[0, 33\)"):
        m._check_tokens(torch.tensor([[0, 33]]))
    with pytest.raises(IndexError):
        m._check_tokens(torch.tensor([[-1, 3]]))
    with pytest.raises(ValueError, match="vocab"):
        _model(vocab=21)
    with pytest.raises(ValueError, match="at most"):
        _model(num_decoder_layers=9)


def test_include_pred_na_N_0_parameter_surface():
    """include_pred_na_N=0 (na_model_utils.py:404-407): edge embedding [128 x 4640]; its 18-atom expansion carries the
    weights at the reference's (a*17+b) pairs and zeros on every pair that involves the absent virtual N_na atom."""
    m = _model(include_pred_na_N=0)
    W = m.features.edge_embedding.weight
    assert tuple(W.shape) == (128, 16 + 16 * 17 * 17)
    with torch.enable_grad():                              # other test modules switch autograd off globally at import
        W18 = m.edge_weight18()
    assert tuple(W18.shape) == (128, spec.EDGE_IN)
    assert torch.equal(W18[:, :16], W[:, :16])
    for a, b in [(0, 0), (3, 16), (16, 5), (16, 16)]:
        assert torch.equal(W18[:, 16 + (a * 18 + b) * 16:16 + (a * 18 + b + 1) * 16], W[:, 16 + (a * 17 + b) * 16:16 + (a * 17 + b + 1) * 16])
    for a, b in [(17, 0), (0, 17), (17, 17), (9, 17)]:
        assert float(W18.detach()[:, 16 + (a * 18 + b) * 16:16 + (a * 18 + b + 1) * 16].abs().max()) == 0.0
    with torch.enable_grad():
        W18.sum().backward()                                   # differentiable: the gradient lands in the [128 x 4640] parameter
    assert W.grad is not None and float(W.grad.min()) == 1.0
    # the default model is untouched
    assert tuple(_model().features.edge_embedding.weight.shape) == (128, spec.EDGE_IN)


def test_bench_respawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks."""
    import importlib.util
    import subprocess
    import sys
    spec_ = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a launcher whose world size contradicts --gpus is refused instead of mislabelling the record
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="contradicts"):
        bench.main()


def test_device_gelu_formula_accuracy():
    """The kernels' GELU (csrc/namp_device.h: x/2 + |x/2| (1 - 2 * 2^-Q(|x|)), Q a degree-6 polynomial, Q(0) = 1) restated in
    float32 numpy with the constants parsed from the header, against the exact erf form in float64: <= 8e-7 absolute over
    [-12, 12] (fp32 rounding of the result at |x| ~ 4) and <= 1e-6 relative near zero; and the linearity the kernels use to move
    the message MLP's last layer behind the K-sum: sum_k w_k (W3 a_k + b3) = W3 (sum_k w_k a_k) + b3 sum_k w_k."""
    from math import erf, sqrt
    src = open(os.path.join(ROOT, "na_mpnn_amd", "csrc", "namp_device.h")).read()
    c = {m.group(1): np.float32(m.group(2)) for m in re.finditer(r"#define NAMP_GELU_(LIM|C[1-6]) (-?[0-9.e+-]+)f", src)}
    assert set(c) == {"LIM", "C1", "C2", "C3", "C4", "C5", "C6"}
    xs = np.linspace(-12.0, 12.0, 480001)
    x = xs.astype(np.float32)
    a = np.abs(np.clip(x, -c["LIM"], c["LIM"]))
    q = c["C6"] * a + c["C5"]
    for k in ("C4", "C3", "C2", "C1"):
        q = q * a + c[k]
    q = q * a + np.float32(1.0)
    e = np.exp2(-q).astype(np.float32)
    h = np.float32(0.5) * x
    y = (h + np.abs(h) * (np.float32(1.0) - np.float32(2.0) * e)).astype(np.float64)
    ref = np.array([0.5 * v * (1.0 + erf(v / sqrt(2.0))) for v in xs])
    assert np.abs(y - ref).max() <= 8e-7
    near0 = (np.abs(xs) < 1e-2) & (xs != 0)
    assert (np.abs(y - ref)[near0] / np.abs(ref[near0])).max() <= 1e-6
    # linearity behind the K-sum
    rng = np.random.default_rng(3)
    W3, b3 = rng.standard_normal((128, 128)), rng.standard_normal(128)
    a2, w = rng.standard_normal((48, 128)), rng.integers(0, 2, 48) / 30.0
    per_edge = ((a2 @ W3.T + b3) * w[:, None]).sum(0)
    hoisted = W3 @ (a2 * w[:, None]).sum(0) + b3 * w.sum()
    assert np.abs(per_edge - hoisted).max() < 1e-12


def test_device_gelu_bf16_mode_polynomial():
    """The bf16 throughput mode's GELU (csrc/namp_device.h: gelu4_bf16mode, x * clamp01(1/2 + x Q(x^2)), Q of degree 4 since round 4)
    restated in float32 numpy with the shipped branch's constants parsed from the header: <= 1.4e-3 absolute over [-14, 14] (a third of
    the bf16 rounding step of its result at |y| >= 1), saturating correctly beyond the fit range, exact at 0."""
    from math import erf, sqrt
    src = open(os.path.join(ROOT, "na_mpnn_amd", "csrc", "namp_device.h")).read()
    c = {k: np.float32(v) for k, v in re.findall(r"#define NAMP_GELU4_(Q[0-4]) (-?[0-9.e+-]+)f", src)}
    assert sorted(c) == ["Q0", "Q1", "Q2", "Q3", "Q4"]
    lead, rest = c["Q4"], [c["Q3"], c["Q2"], c["Q1"], c["Q0"]]
    # the training backward's gelu / gelu' pair (namp_train_dw.h) evaluates the SAME polynomial: no second copy of the constants
    dw = open(os.path.join(ROOT, "na_mpnn_amd", "csrc", "namp_train_dw.h")).read()
    fn = dw[dw.index("f4 dw_gelu_split4_bf16("):]
    fn = fn[:fn.index("\n}\n")]
    assert "NAMP_GELU4_Q4" in fn and "NAMP_GELU4_Q0" in fn and not re.search(r"[0-9]e-0[0-9]f", fn.split("fmed3f")[0])
    xs = np.linspace(-14.0, 14.0, 560001)
    x = xs.astype(np.float32)
    t = x * x
    q = np.full_like(x, lead)
    for c in rest:
        q = q * t + c
    y = (x * np.clip(x * q + np.float32(0.5), 0, 1)).astype(np.float64)
    ref = np.array([0.5 * v * (1.0 + erf(v / sqrt(2.0))) for v in xs])
    assert np.abs(y - ref).max() <= 1.4e-3
    far = np.abs(xs) > 6
    assert np.abs(y - ref)[far].max() <= 1e-6            # saturated: x or 0
    assert y[len(xs) // 2] == 0.0


def _levels_ref(E_idx, order, gf, gl):
    """sample_levels_kernel (csrc/namp_kernels.h) restated: one level per symmetry group = 1 + the highest level among its members'
    earlier neighbours in EARLIER groups."""
    L = len(order)
    rank = np.empty(L, dtype=np.int64); rank[order] = np.arange(L)
    lv = np.full(L, -1); out = np.zeros(L, dtype=np.int32)
    dg = -1
    for t in range(L):
        i, vf = order[t], gf[t]
        d = max([lv[j] for j in E_idx[i] if rank[j] < vf], default=-1)
        dg = d if vf == t else max(dg, d)
        if gl[t]:
            for v in range(vf, t + 1):
                lv[order[v]] = dg + 1; out[v] = dg + 1
    return out, rank


@pytest.mark.parametrize("split", [False, True])
def test_level_work_lists_of_the_symmetric_sampler(split):
    """Host side of the level-parallel sampler (model.level_work_lists): every visit belongs to exactly one work item, items come sorted by
    level, a symmetry group is one item unless `split` and none of its members is a graph neighbour of another member, every earlier
    neighbour outside the group sits in a lower level, and the deferred-draw lists name every group once per stream."""
    from na_mpnn_amd.model import level_work_lists
    rng = np.random.default_rng(5)
    L, K, B_dec = 60, 8, 3
    E_idx = np.stack([np.concatenate(([i], rng.choice(np.delete(np.arange(L), i), K - 1, replace=False))) for i in range(L)])
    groups_in = [[3, 4, 5], [10, 40], [20, 50, 7, 33], [11, 12]]
    E_idx[10, 1:] = [x for x in range(L) if x not in (10, 40)][:K - 1]                    # group [10, 40]: no internal edge
    E_idx[40, 1:] = [x for x in range(L) if x not in (10, 40)][5:5 + K - 1]
    E_idx[4, 1] = 3                                                                       # group [3, 4, 5]: 4 neighbours 3
    for a in (20, 50, 7, 33):
        E_idx[a, 1:] = [x for x in range(L) if x not in (20, 50, 7, 33)][a % 9:a % 9 + K - 1]
    E_idx[12, 2] = 11
    group_of = {r: g for g in groups_in for r in g}
    seen, groups = set(), []
    for r in rng.permutation(L).tolist():
        if r not in seen:
            groups.append(list(group_of.get(r, (r,)))); seen.update(groups[-1])
    order = np.array([r for g in groups for r in g])
    gf, gl, v = [], [], 0
    for g in groups:
        gf += [v] * len(g); gl += [0] * (len(g) - 1) + [1]; v += len(g)
    lev, rank = _levels_ref(E_idx, order, gf, gl)
    T = lambda a, dt=torch.int32: torch.tensor(np.asarray(a)).to(dt)
    level = T(lev).repeat(B_dec, 1)
    sel, flat, work_n, close, close_off = level_work_lists(level, T(gf).repeat(B_dec, 1), T(gl).repeat(B_dec, 1), T(order, torch.int64),
                                                           T(E_idx, torch.int64), split=split)
    assert torch.equal(flat, flat.sort().values) and int(work_n.sum()) == B_dec * L
    covered = np.zeros((B_dec, L), dtype=int)
    for s_, n_, f_ in zip(sel.tolist(), work_n.tolist(), flat.tolist()):
        b, v0 = divmod(s_, L)
        covered[b, v0:v0 + n_] += 1
        assert gf[v0] == gf[v0 + n_ - 1] and f_ == lev[v0]                                 # one group, the group's level
        members = order[gf[v0]:gf[v0] + gf.count(gf[v0])]
        internal = any(j in members and j != i for i in members for j in E_idx[i])
        assert n_ == (len(members) if (internal or not split) else 1), (members, internal, n_)
        for i in order[v0:v0 + n_]:                                                        # dependencies sit in lower levels
            for j in E_idx[i]:
                if rank[j] < gf[v0]:
                    assert lev[rank[j]] < f_
    assert (covered == 1).all()
    if not split:
        assert close is None and close_off is None
        assert len(sel) == B_dec * len(groups)
    else:
        assert len(sel) > B_dec * len(groups)                                              # [10, 40] and [20, 50, 7, 33] were split
        assert close.shape == (B_dec * len(groups), 2)
        lv_c = [lev[v_] for _, v_ in close.tolist()]
        assert lv_c == sorted(lv_c) and all(gl[v_] for _, v_ in close.tolist())
        assert sorted(map(tuple, close.tolist())) == sorted((b, v_) for b in range(B_dec) for v_ in range(L) if gl[v_])
        off = close_off.tolist()
        for l_ in range(max(lev) + 1):
            assert [lv_c[q] for q in range(off[l_], off[l_ + 1])] == [l_] * (off[l_ + 1] - off[l_])
        assert off[max(lev) + 1] == len(lv_c)
    # without groups: one item per visit
    sel1, flat1, wn1, c1, co1 = level_work_lists(level, None, None, T(order, torch.int64), T(E_idx, torch.int64), split=False)
    assert wn1 is None and c1 is None and sel1.numel() == B_dec * L and torch.equal(flat1, flat1.sort().values)
