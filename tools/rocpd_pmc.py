#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 (rocpd sqlite) run.

    python tools/rocpd_pmc.py gpurun_out/pmc/k_results.db [kernel-substring ...]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name)[:60]


def main(path, filters):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection").fetchall()
    per = defaultdict(lambda: defaultdict(float))     # (kernel, dispatch) -> counter -> summed value
    dur = {}
    for k, d, c, v, du in rows:
        k = short(k)
        if filters and not any(f in k for f in filters):
            continue
        per[(k, d)][c] += v
        dur[(k, d)] = du
    agg = defaultdict(lambda: defaultdict(list))
    for (k, d), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
        agg[k]["duration_us"].append(dur[(k, d)] / 1e3)
    for k, cs in agg.items():
        n = len(cs["duration_us"])
        print(f"== {k}  ({n} dispatches)")
        for c in sorted(cs):
            vals = cs[c]
            print(f"   {c:32s} {sum(vals) / len(vals):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
