"""Symmetry-tied ProteinMPNN.sample() on cuda:0: decoded by dependency level (groups as work items) against the sequential walk.
    python tools/sample_sym_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
w = synth.make_weights(0)
for n, copies, k, bs in ((1000, 2, 48, 1), (1000, 2, 48, 8), (600, 3, 32, 4), (400, 4, 32, 1)):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                    polytype_to_int=spec.polytype_to_int())
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()}); m = m.to(dev).eval()
    cx = synth.make_complex(seed=3, n=n, n_chains=copies)
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k_, v in cx.items()}
    per = n // copies
    groups = [[i + c * per for c in range(copies)] for i in range(per)]
    fd.update({"batch_size": bs, "temperature": 0.1, "bias": torch.zeros(1, n, 33, device=dev), "symmetry_residues": groups,
               "symmetry_weights": [[1.0 / copies] * copies for _ in groups], "randn": torch.randn(bs, n, device=dev)})
    V, E, E_idx = m.featurize(fd)
    m.featurize = lambda fd_, _r=(V, E, E_idx): _r            # time the sampler, not the featuriser
    res = {}
    for name, lvl, walk in (("persistent level walk", True, True), ("one launch per level", True, False), ("sequential walk", False, False)):
        m.sample_level_parallel, m.sample_level_walk = lvl, walk
        torch.manual_seed(1); out = m.sample(fd); torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps): m.sample(fd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        res[name] = (dt, out)
        print(f"N={n} ({copies} copies of {per}) K={k} batch_size={bs}: {name}: {dt*1e3:.1f} ms"
              + (f", {int(out['levels'])} levels for {per} groups" if "levels" in out else ""), flush=True)
    a, b = res["persistent level walk"][1], res["sequential walk"][1]
    print("   identical draws and log-probs:", bool(torch.equal(a["S"], b["S"]) and torch.equal(a["log_probs"], b["log_probs"])))
