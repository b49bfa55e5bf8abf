"""Upper bound of folding the decoding-order sort into another launch: score() from coordinates with the sort replaced by cached results, alternating."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
m = bench._feat_model(dev)
fd = bench._feat_inputs(dev, "cfg2"); fd["batch_size"] = 1
fd["randn"] = torch.randn(tuple(fd["mask"].shape), generator=torch.Generator().manual_seed(7)).to(dev)
real = m.order_and_rank
o, r = real(fd["mask"], fd["chain_mask"], fd["randn"]); torch.cuda.synchronize()
def cached(mask, chain_mask, randn, defer=False):
    return o, r
for rep in range(3):
    for name, fn in (("sort on the side stream", real), ("no sort (cached)", cached)):
        m.order_and_rank = fn
        for _ in range(10):
            m.score(fd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            m.score(fd)
        torch.cuda.synchronize()
        print(f"{name}: score() {1e3 * (time.perf_counter() - t0) / 200:.4f} ms", flush=True)
