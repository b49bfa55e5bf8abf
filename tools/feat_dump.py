"""Dump E / h_E of the cfg2 complex (x3 products, one launch and in parts) to a file, or compare two dumps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f"{k}: max|d| = {d.max():.3e}, mean|d| = {d.mean():.3e}, differing = {(d > 0).mean():.4f}, max|a| = {np.abs(a[k]).max():.3f}")
    sys.exit(0)
import bench
from na_mpnn_amd import hip
dev = torch.device("cuda:0")
m = bench._feat_model(dev)
fd = bench._feat_inputs(dev, "cfg2"); fd["batch_size"] = 1
out = {}
for prec in ("x3", "fp32"):
    m.message_precision = prec
    for mask in (11, 43):
        hip.lib().namp_set_bf16p(mask)
        _, E, hE, I = m._featurize_hip(fd, want_E=True, want_hE=True)
        out[f"E_{prec}_{mask}"] = E.cpu().numpy(); out[f"hE_{prec}_{mask}"] = hE.cpu().numpy()
np.savez(sys.argv[1], **out)
