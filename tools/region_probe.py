"""Where does a 20-step timed region of cfg2 spend its time?  Run from the root of a tree (HEAD or tools/_variants/r3tree):

    python /root/repo/tools/region_probe.py [fp32|x3] [tag]

For the driver-style region (50 ms pre-spin, 5 warm-up steps, sync, K steps, sync) it prints
  * the wall-clock ms/step of 12 repetitions at K = 20 and at K = 200 / 1000,
  * per-step device times of a 20-step region from events recorded behind every step (first step, median step, last),
  * the host time to enqueue the 20 steps and the time from the last kernel's end to synchronize() returning,
  * the same region with variations: no pre-spin, pre-spin without per-step sync, gc.collect() right before, a 2 ms idle gap.
"""
import gc
import json
import statistics
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
tag = sys.argv[2] if len(sys.argv) > 2 else "tree"
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
r = bench.Runner(dev, 1, 1000, 48, seed=3, precision=prec)
sync = torch.cuda.synchronize


def region(K, prespin="sync", collect=False, idle_ms=0.0, warm=5):
    if prespin:
        t = time.perf_counter()
        while time.perf_counter() - t < 0.05:
            r.step()
            if prespin == "sync":
                sync()
    for _ in range(warm):
        r.step()
    sync()
    if collect:
        gc.collect()
    if idle_ms:
        time.sleep(idle_ms * 1e-3)
    t0 = time.perf_counter()
    for _ in range(K):
        r.step()
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    return (t2 - t0) / K * 1e3, (t1 - t0) * 1e3


def stats(xs):
    xs = sorted(xs)
    return {"min": round(xs[0], 4), "med": round(statistics.median(xs), 4), "max": round(xs[-1], 4)}


out = {"tag": tag, "precision": prec}
region(20)
gc.collect()
gc.disable()
for K in (20, 50, 200, 1000):
    out[f"wall_K{K}"] = stats([region(K)[0] for _ in range(12 if K <= 50 else 4)])
out["enqueue_ms_K20"] = stats([region(20)[1] for _ in range(12)])
out["wall_K20_noprespin"] = stats([region(20, prespin=None)[0] for _ in range(12)])
out["wall_K20_prespin_nosync"] = stats([region(20, prespin="nosync")[0] for _ in range(12)])
out["wall_K20_collect"] = stats([region(20, collect=True)[0] for _ in range(12)])
out["wall_K20_idle2ms"] = stats([region(20, idle_ms=2.0)[0] for _ in range(12)])
out["wall_K20_warm50"] = stats([region(20, warm=50)[0] for _ in range(12)])

# per-step device times
per_first, per_med, per_last, tail_us, tot = [], [], [], [], []
for _ in range(12):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        r.step()
        sync()
    for _ in range(5):
        r.step()
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(20):
        r.step()
        ev[i + 1].record()
    sync()
    t2 = time.perf_counter()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
    per_first.append(d[0]); per_med.append(statistics.median(d)); per_last.append(d[-1])
    tot.append(ev[0].elapsed_time(ev[20]) / 20)
    tail_us.append(((t2 - t0) * 1e3 - ev[0].elapsed_time(ev[20])) * 1e3)
out["dev_first_step_ms"] = stats(per_first)
out["dev_median_step_ms"] = stats(per_med)
out["dev_last_step_ms"] = stats(per_last)
out["dev_region_per_step_ms"] = stats(tot)
out["wall_minus_device_us"] = stats(tail_us)
print(json.dumps(out))
