#!/usr/bin/env python
"""Device-idle gaps in a rocprofv3 kernel trace (rocpd sqlite): total busy / idle time per step-sized window and the largest gaps with
the kernels on either side — where does a launch-heavy step leave the GPU waiting for the host?

    python tools/rocpd_gaps.py <results.db> [--top 25] [--min-gap-us 15]
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--min-gap-us", type=float, default=15.0)
    ap.add_argument("--last-ms", type=float, default=0.0, help="analyse only the last so many milliseconds of the trace (steady state)")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = sorted(db.execute(f"select {ncol}, start, end from kernels").fetchall(), key=lambda r: r[1])
    t1 = max(r[2] for r in rows)
    if a.last_ms > 0:
        rows = [r for r in rows if r[1] >= t1 - a.last_ms * 1e6]
    t0 = rows[0][1]
    busy = 0.0
    cur_end = rows[0][1]
    gaps = []
    for i, (n, s, e) in enumerate(rows):
        if s > cur_end:
            gaps.append(((s - cur_end) / 1e3, short(rows[i - 1][0]), short(n), (cur_end - t0) / 1e6))
        if e > cur_end:
            busy += (e - max(s, cur_end)) / 1e3
            cur_end = e
    total = (t1 - t0) / 1e3
    print(f"{len(rows)} kernels over {total / 1e3:.1f} ms: busy {busy / 1e3:.1f} ms ({100 * busy / total:.1f} %), idle {(total - busy) / 1e3:.1f} ms")
    big = [g for g in gaps if g[0] >= a.min_gap_us]
    print(f"{len(gaps)} gaps, {len(big)} of them >= {a.min_gap_us} us holding {sum(g[0] for g in big) / 1e3:.2f} ms; small gaps hold {sum(g[0] for g in gaps if g[0] < a.min_gap_us) / 1e3:.2f} ms")
    print("| gap us | at ms | after | before |")
    print("|---:|---:|---|---|")
    for g in sorted(big, key=lambda g: -g[0])[:a.top]:
        print(f"| {g[0]:.0f} | {g[3]:.1f} | `{g[1]}` | `{g[2]}` |")
    # by neighbour kernel: which kernel is most often followed by a gap
    agg = {}
    for g in gaps:
        k = g[1]
        agg.setdefault(k, [0, 0.0])
        agg[k][0] += 1; agg[k][1] += g[0]
    print("\n| kernel followed by idle | gaps | total idle us |")
    print("|---|---:|---:|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f"| `{k}` | {v[0]} | {v[1]:.0f} |")


if __name__ == "__main__":
    main()
