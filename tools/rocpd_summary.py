#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into a per-kernel stats table.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name if len(name) < 90 else name[:87] + "..."


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {ncol}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
