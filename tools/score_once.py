"""score() from coordinates on the cfg2 complex, a few calls (for a kernel trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
m = bench._feat_model(dev)
fd = bench._feat_inputs(dev, "cfg2"); fd["batch_size"] = 1
fd["randn"] = torch.randn(tuple(fd["mask"].shape), generator=torch.Generator().manual_seed(7)).to(dev)
for _ in range(12):
    m.score(fd)
torch.cuda.synchronize()
