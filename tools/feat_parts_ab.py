"""namp_featurize (h_E only, as score() calls it) in one launch against two / three / four parts, alternating on one box: cfg2 (N = 1000, K = 48) and the
97-residue chain of cfg1 (K = 32)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from na_mpnn_amd import hip, synth
dev = torch.device("cuda:0")
L = hip.lib()
base = 11
MASKS = [int(x) for x in os.environ.get('FEAT_MASKS', '11,43,107,171').split(',')]
cases = [("cfg2 N=1000 K=48", bench._feat_model(dev), bench._feat_inputs(dev, "cfg2"))]
cx = synth.make_complex(seed=5, n=97, n_chains=1, frac_protein=1.0, frac_dna=0.0)
fd97 = {k: torch.from_numpy(v)[None].to(dev) for k, v in cx.items() if hasattr(v, "shape")}
cases.append(("protein N=97", cases[0][1], fd97))
for name, m, fd in cases:
    fd["batch_size"] = 1
    for rep in range(2):
        for mask in MASKS:
            L.namp_set_bf16p(mask)
            for _ in range(10):
                m._featurize_hip(fd, want_E=False, want_hE=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                m._featurize_hip(fd, want_E=False, want_hE=True)
            torch.cuda.synchronize()
            print(f"{name} mask {mask}: featurize {1e6 * (time.perf_counter() - t0) / 200:.1f} us", flush=True)
