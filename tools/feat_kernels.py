"""Per-kernel device time of one featurisation (prep_atoms, knn, edge_features) at N residues, via the library's own
HIP-event profile hooks is too coarse (one kind) — so time featurize() with and without pieces using torch events.
    NAMP_LIB_PATH=... python tools/feat_kernels.py [N ...]"""
import os, sys
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
tag = os.path.basename(os.environ.get("NAMP_LIB_PATH", "default"))
for n in [int(a) for a in sys.argv[1:]] or [1000]:
    cx = synth.make_complex(seed=3, n=n)
    fd = {k: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k, v in cx.items()}; fd["batch_size"] = 1
    for _ in range(3): m.featurize(fd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): m.featurize(fd)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag} N={n} featurize {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
