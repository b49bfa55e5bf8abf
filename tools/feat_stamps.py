"""Per-workgroup time line of the edge-feature launch of one complex (build with -DFEAT_STAMPS; NAMP_LIB_PATH selects it): entry / end of set-up / end
of every workgroup on s_memrealtime (100 MHz, chip-wide), the residue block and part it took, the chunks it walked, its XCD."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from na_mpnn_amd import hip
L = hip.lib()
dbg = C.CDLL(os.environ["NAMP_LIB_PATH"]).namp_debug_feat_stamps
dev = torch.device("cuda:0")
m = bench._feat_model(dev)
fd = bench._feat_inputs(dev, "cfg2"); fd["batch_size"] = 1
TICK = 100.0
for mask in (11, 43, 107, 171):
    L.namp_set_bf16p(mask)
    for _ in range(20):
        m._featurize_hip(fd, want_E=False, want_hE=True)
    torch.cuda.synchronize()
    buf = np.zeros((1024, 8), dtype=np.uint64)
    dbg(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    b = buf.astype(np.int64)
    b = b[b[:, 2] > b[:, 2].max() - 1000 * TICK]            # the last launch's rows (older, larger launches leave stale ones)
    t0 = b[:, 0].min()
    start, end, setup = (b[:, 0] - t0) / TICK, (b[:, 2] - t0) / TICK, (b[:, 1] - b[:, 0]) / TICK
    dur = end - start
    ch = b[:, 5]; parts = b[:, 4] >> 8; xcc = b[:, 7] & 15
    print(f"== mask {mask}: {len(b)} workgroups ran ({int(np.sum(parts > 1))} on split blocks); first entry to last end {end.max():.1f} us; "
          f"entries within 2 us: {int(np.sum(start < 2.0))}, within 10 us: {int(np.sum(start < 10.0))}, last entry {start.max():.1f} us")
    print(f"   set-up {setup.mean():.2f} us (max {setup.max():.2f});  chunks per workgroup: min {ch.min()} mean {ch.mean():.1f} max {ch.max()};  "
          f"sum of durations / 256 CUs = {dur.sum() / 256:.1f} us")
    for lo, hi in ((0, 12), (12, 20), (20, 30), (30, 60)):
        sel = (ch >= lo) & (ch < hi)
        if sel.any():
            first = sel & (start < 10.0); later = sel & (start >= 10.0)
            f = lambda q: f"{int(q.sum())} wgs, {dur[q].mean():.1f} us, {((dur[q] - setup[q]) / np.maximum(ch[q], 1)).mean():.2f} us/chunk" if q.any() else "-"
            print(f"   {lo:2d}-{hi - 1:2d} chunks: entered in the first 10 us: {f(first)};  later: {f(later)}")
    fin = np.argsort(end)[-5:]
    print("   last finishers: " + ", ".join(f"(blk {b[i,3]}, part {b[i,4] & 255}/{parts[i]}, {ch[i]} ch, xcd {xcc[i]}, {start[i]:.0f}..{end[i]:.0f})" for i in fin))
    edges = np.arange(0, end.max() + 10, 10.0)
    busy = [(int(np.sum((start < e + 10) & (end > e)))) for e in edges[:-1]]
    print("   workgroups alive per 10 us: " + " ".join(str(x) for x in busy))
    print("   per XCD last end: " + " ".join(f"{end[xcc == x].max():.0f}" for x in range(8) if (xcc == x).any()))
