"""Time ProteinMPNN.sample() (persistent HIP sampler) on cuda:0; optionally the CPU oracle for comparison."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import spec, synth
from na_mpnn_amd.model import ProteinMPNN
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
w = synth.make_weights(0)
def model(k):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                    polytype_to_int=spec.polytype_to_int())
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()}); return m.to(dev).eval()
for n, k, bs in ((1000, 48, 1), (1000, 32, 1), (400, 32, 30), (400, 32, 4), (97, 32, 1)):
    cx = synth.make_complex(seed=3, n=n)
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k_, v in cx.items()}
    fd.update({"batch_size": bs, "temperature": 0.1, "bias": torch.zeros(1, n, 33, device=dev), "symmetry_residues": [[]],
               "symmetry_weights": [[]], "randn": torch.randn(bs, n, device=dev)})
    m = model(k)
    m.sample_level_parallel = "--sequential" not in sys.argv
    V, E, E_idx = m.featurize(fd)
    m.featurize = lambda fd_, _r=(V, E, E_idx): _r            # time the sampler, not the (torch-op) featuriser
    m.sample(fd); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): m.sample(fd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"N={n} K={k} batch_size={bs}: sample() {dt*1e3:.1f} ms = {dt/n*1e6:.1f} us/step, {bs*n/dt:.0f} sampled residues/s", flush=True)
if "--cpu" in sys.argv:
    from oracle import cpu_ref
    torch.set_num_threads(8)
    for n, k, bs in ((400, 32, 4), (97, 32, 1)):
        cx = synth.make_complex(seed=3, n=n)
        fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None] for k_, v in cx.items()}
        fd.update({"batch_size": bs, "temperature": 0.1, "bias": torch.zeros(1, n, 33), "symmetry_residues": [[]],
                   "symmetry_weights": [[]], "randn": torch.randn(bs, n)})
        wt = {k_: torch.from_numpy(v) for k_, v in w.items()}
        t0 = time.perf_counter(); cpu_ref.sample(wt, fd, k); dt = time.perf_counter() - t0
        print(f"CPU oracle (8 threads, incl. features) N={n} K={k} batch_size={bs}: {dt*1e3:.0f} ms, {bs*n/dt:.0f} sampled residues/s", flush=True)
