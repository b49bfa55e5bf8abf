"""Time the training step (fwd + bwd + Adam) at a cfg5-like shape and list the top device kernels.

    python tools/train_time.py [--B 16] [--N 1500] [--K 48] [--steps 5] [--profile]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import spec, synth, train          # noqa: E402
from na_mpnn_amd.model import ProteinMPNN           # noqa: E402


def make_batch(B, N, dev, seed=5):
    cxs = [synth.make_complex(seed=seed + b, n=N, n_chains=4) for b in range(B)]
    fd = {k: torch.from_numpy(np.stack([c[k] for c in cxs])).to(dev) for k in cxs[0]}
    fd["S"] = fd["S"].long()
    return fd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16); ap.add_argument("--N", type=int, default=1500)
    ap.add_argument("--K", type=int, default=48); ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.1); ap.add_argument("--profile", action="store_true")
    ap.add_argument("--precision", default=None)
    ap.add_argument("--shapes", action="store_true")
    ap.add_argument("--mem", action="store_true", help="allocated / peak bytes around every autograd Function of the step")
    ap.add_argument("--each", default=None, help="print every launch duration (us) of the device kernels whose name contains this")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rti = spec.restype_to_int()
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=a.K, dropout=a.dropout, atom_dict=spec.atom_dict(),
                    restype_to_int=rti, polytype_to_int=spec.polytype_to_int(), augment_eps=0.1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0).items()})
    m.to(dev).train()
    if a.precision:
        m.message_precision = a.precision
    fd = make_batch(a.B, a.N, dev)
    opt = train.get_std_opt(m.parameters(), 128, 0)
    rm, rn = train.polymer_restype_tables(rti, 33, dev)
    no_loss = torch.tensor([rti[t] for t in ("UNK", "DX", "RX", "MAS", "PAD")], device=dev)
    step = lambda: train.train_step(m, opt, fd, rm, rn, no_loss, loss_tokens=6000.0, gradient_norm=1.0)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # host side: how long does Python take to ENQUEUE a step (no sync inside), against the step's device time?
    th = []
    for _ in range(a.steps):
        torch.cuda.synchronize()
        t1 = time.perf_counter(); step(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        th.append((t2 - t1, t3 - t1))
    print("host enqueue / step total (ms), device idle at the start of each:", ", ".join(f"{x * 1e3:.1f} / {y * 1e3:.1f}" for x, y in th))
    print(f"B={a.B} N={a.N} K={a.K}: {dt * 1e3:.1f} ms/step, {a.B * a.N / dt:.0f} residues/s trained, loss {float(loss):.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if a.mem:
        log = []

        def wrap(cls, which):
            fn = getattr(cls, which)

            def w(*args, **kw):
                before = torch.cuda.memory_allocated(); torch.cuda.reset_peak_memory_stats()
                out = fn(*args, **kw)
                log.append((f"{cls.__name__}.{which}", before, torch.cuda.max_memory_allocated(), torch.cuda.memory_allocated()))
                return out
            setattr(cls, which, staticmethod(w))
        for cls in vars(train).values():
            if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
                wrap(cls, "forward"); wrap(cls, "backward")
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        step(); torch.cuda.synchronize()
        G = 2.0 ** 30
        print("== memory around every autograd Function (GiB): allocated before, peak inside, allocated after")
        for name, b, pk, af in log:
            print(f"{name:34s} {b / G:6.2f} {pk / G:6.2f} {af / G:6.2f}")
        print("max peak:", max(log, key=lambda r: r[2]))
    if a.profile:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes="--shapes" in sys.argv) as prof:
            step(); torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
        from torch.autograd import DeviceType
        if "--shapes" in sys.argv:                        # stock (aten::) ops by input shape, sorted by the device time of their own kernels
            rows = [(e.key, str(e.input_shapes)[:110], e.count, e.self_device_time_total) for e in prof.key_averages(group_by_input_shape=True)
                    if e.key.startswith("aten::") and e.self_device_time_total > 0]
            rows.sort(key=lambda r: -r[3])
            print("== stock ops by input shape (self device time)")
            for k, sh, c, t in rows[:45]:
                print(f"{t:9.1f} us {c:4d} x  {k:28s} {sh}")
        if a.each:
            ds = [round(e.device_time, 1) for e in prof.events() if e.device_type == DeviceType.CUDA and a.each in e.name]
            print(f"== {a.each}: {len(ds)} launches, us each:", ds)
        # device kernels only, all of them: name, calls, total us
        rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_type == DeviceType.CUDA]
        rows.sort(key=lambda r: -r[2])
        tot = sum(r[2] for r in rows)
        print(f"== device kernels: {len(rows)} kinds, {sum(r[1] for r in rows)} launches, {tot / 1e3:.2f} ms")
        for k, c, t in rows:
            print(f"{t:10.1f} us {c:5d} x  {k[:150]}")


if __name__ == "__main__":
    main()
