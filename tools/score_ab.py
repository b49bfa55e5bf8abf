"""score() from coordinates (N = 1000, K = 48; and a 13 x ~2,400 batch): the decoding-order sort on a side stream vs in the calling stream, alternating."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
m = bench._feat_model(dev)
for which, n in (("cfg2", 40), ("cfg4", 10)):
    fd = bench._feat_inputs(dev, which)
    fd["batch_size"] = 1
    fd["randn"] = torch.randn(tuple(fd["mask"].shape), generator=torch.Generator().manual_seed(7)).to(dev)
    for rep in range(3):
        for side in (False, True):
            m.order_side_stream = side
            for _ in range(5):
                m.score(fd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                m.score(fd)
            torch.cuda.synchronize()
            print(f"{which} side_stream={side}: score() {1e3 * (time.perf_counter() - t0) / n:.4f} ms", flush=True)
