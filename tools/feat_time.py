"""Time the current (torch-ops) featuriser and the full score() from coordinates on cuda:0."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
w = synth.make_weights(0)
m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=48, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                polytype_to_int=spec.polytype_to_int())
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval()
cx = synth.make_complex(seed=3, n=1000)
fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k, v in cx.items()}; fd["batch_size"] = 1
Es = {}
for prec in ("fp32", "x3"):
    m.message_precision = prec
    Es[prec] = m.featurize(fd)[1].clone()
    lp = m.score(fd)["log_probs"].clone()
    Es[prec + "_lp"] = lp
print("featuriser x3 vs fp32: max |dE| = %.3g (max |E| %.3g), max |dlogp| = %.3g" % (
    float((Es["x3"] - Es["fp32"]).abs().max()), float(Es["fp32"].abs().max()), float((Es["x3_lp"] - Es["fp32_lp"]).abs().max())))
for n in (1000,):
    cx = synth.make_complex(seed=3, n=n)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k, v in cx.items()}; fd["batch_size"] = 1
    for name, f in (("featurize", lambda: m.featurize(fd)), ("score", lambda: m.score(fd))):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"N={n} {name}: {dt*1e3:.3f} ms  ({n/dt:.0f} residues/s)")
