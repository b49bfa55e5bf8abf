"""edge_features_kernel with its split-bf16 steps software-pipelined against the build without (-DFEAT_NOPIPE, NAMP_LIB_PATH): run once per library;
prints featurize timings at cfg2 (one complex) and the cfg4 batch, and a checksum of E / h_E (bit-identity across the two builds)."""
import os, sys, time, hashlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
m = bench._feat_model(dev)
for which, n in (("cfg2", 200), ("cfg4", 30)):
    fd = bench._feat_inputs(dev, which); fd["batch_size"] = 1
    for prec in ("x3", "bf16"):
        m.message_precision = prec
        _, E, hE, I = m._featurize_hip(fd, want_E=True, want_hE=True)
        torch.cuda.synchronize()
        dg = hashlib.sha1(E.cpu().numpy().tobytes() + hE.cpu().numpy().tobytes()).hexdigest()[:12]
        for rep in range(2):
            for _ in range(5):
                m._featurize_hip(fd, want_E=False, want_hE=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                m._featurize_hip(fd, want_E=False, want_hE=True)
            torch.cuda.synchronize()
            print(f"{which} {prec}: featurize {1e6 * (time.perf_counter() - t0) / n:.1f} us   sha1(E, h_E) {dg}", flush=True)
