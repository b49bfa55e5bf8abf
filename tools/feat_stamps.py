"""Per-workgroup time line of the edge-feature launch of one complex (build with -DFEAT_STAMPS; NAMP_LIB_PATH selects it): entry / end of set-up / end
of every workgroup (s_memtime: one counter per XCD, so times are taken relative to the XCD's first entry), the residue block and part it took, the
chunks it walked."""
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
TICK = float(os.environ.get("TICKS_PER_US", "2400"))    # s_memtime follows the shader clock here (it slows down when the chip idles)
for mask in (11, 43, 107, 171):
    L.namp_set_bf16p(mask)
    for _ in range(5):
        m._featurize_hip(fd, want_E=False, want_hE=True)
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    for _ in range(100):
        m._featurize_hip(fd, want_E=False, want_hE=True)
    torch.cuda.synchronize()
    wall = 1e4 * (time.perf_counter() - t_w)
    buf = np.zeros((1024, 8), dtype=np.uint64)
    dbg(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    n = {11: 1, 43: 2, 107: 3, 171: 4}[mask] * 250          # K = 48: three waves per residue, four residues per workgroup
    b = buf[:n].astype(np.int64)
    b = b[b[:, 2] > 0] if mask == 11 else b[(b[:, 4] >> 8) > 0]
    # stale rows of an earlier, larger launch: keep the rows whose end lies within 1 ms of the newest
    xcc = b[:, 7] & 15
    start = np.zeros(len(b)); end = np.zeros(len(b))
    for x in range(8):
        sel = xcc == x
        if not sel.any(): continue
        newest = b[sel, 2].max()
        t0 = b[sel & (b[:, 2] > newest - 400 * TICK), 0].min()
        start[sel] = (b[sel, 0] - t0) / TICK; end[sel] = (b[sel, 2] - t0) / TICK
    fresh = start >= 0
    b, start, end, xcc = b[fresh], start[fresh], end[fresh], xcc[fresh]
    setup = (b[:, 1] - b[:, 0]) / TICK
    dur = end - start
    ch = b[:, 5]; parts = b[:, 4] >> 8
    print(f"== mask {mask}: {len(b)} workgroups ran ({int(np.sum(parts > 1))} of split blocks); featurize call {wall:.1f} us; span of the launch {end.max():.1f} us; "
          f"starts within 2 us: {int(np.sum(start < 2.0))}, last start {start.max():.1f} us")
    print(f"   set-up {setup.mean():.2f} us (max {setup.max():.2f});  chunks per workgroup: min {ch.min()} mean {ch.mean():.1f} max {ch.max()}")
    body = dur - setup
    A = np.stack([ch, np.ones_like(ch)], 1).astype(np.float64)
    co, *_ = np.linalg.lstsq(A, body, rcond=None)
    print(f"   body = {co[0]:.2f} us x chunks + {co[1]:.2f} us (least squares);  longest workgroup {dur.max():.1f} us with {ch[dur.argmax()]} chunks;  "
          f"sum of workgroup durations / 256 CUs = {dur.sum() / 256:.1f} us")
    fin = np.argsort(end)[-6:]
    print("   last finishers: " + ", ".join(f"(blk {b[i,3]}, part {b[i,4] & 255}/{parts[i]}, {ch[i]} ch, {start[i]:.0f}..{end[i]:.0f})" for i in fin))
    hist, edges = np.histogram(end, bins=10, range=(0, end.max()))
    print("   ends histogram (tenths of the span): " + " ".join(str(h) for h in hist))
    hist, edges = np.histogram(start, bins=10, range=(0, end.max()))
    print("   starts histogram:                    " + " ".join(str(h) for h in hist))
