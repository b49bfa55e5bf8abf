"""Device-to-device copy bandwidth on this box (the practical HBM ceiling for the gather kernel: SURVEY 8(d))."""
import torch
dev = torch.device("cuda:0")
for mb in (256, 1024, 3072):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev).normal_(); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy {mb} MiB: {ms:.3f} ms  read+write {2 * n * 4 / ms / 1e6:.0f} GB/s")
    e0.record()
    for _ in range(10): b.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fill {mb} MiB: {ms:.3f} ms  write {n * 4 / ms / 1e6:.0f} GB/s")
    e0.record()
    for _ in range(10): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"sum  {mb} MiB: {ms:.3f} ms  read {n * 4 / ms / 1e6:.0f} GB/s")
