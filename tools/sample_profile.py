"""Device kernels of ONE design call (model.sample on a 97-residue chain, K = 32, batch_size 1 — bench.py's cfg1) and the host-side time line."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
from torch.profiler import profile, ProfilerActivity
from torch.autograd import DeviceType
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
torch.manual_seed(0)
w = synth.make_weights(0)
n, k, bs = (int(sys.argv[1]) if len(sys.argv) > 1 else 97), 32, 1
m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                polytype_to_int=spec.polytype_to_int())
m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()}); m = m.to(dev).eval()
cx = synth.make_complex(seed=3, n=n)
fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k_, v in cx.items()}
fd.update({"batch_size": bs, "temperature": 0.1, "bias": torch.zeros(1, n, 33, device=dev), "symmetry_residues": [[]],
           "symmetry_weights": [[]], "randn": torch.randn(bs, n, device=dev)})
for _ in range(3):
    out = m.sample(fd)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = m.sample(fd); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("host enqueue / total ms per call:", ", ".join(f"{a:.2f}/{b:.2f}" for a, b in ts), "levels", int(out["levels"]))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    out = m.sample(fd); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_type == DeviceType.CUDA]
rows.sort(key=lambda r: -r[2])
print(f"== device kernels: {len(rows)} kinds, {sum(r[1] for r in rows)} launches, {sum(r[2] for r in rows) / 1e3:.3f} ms")
for key, c, t in rows[:40]:
    print(f"{t:10.1f} us {c:5d} x  {key[:140]}")
