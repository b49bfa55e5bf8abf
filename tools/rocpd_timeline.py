#!/usr/bin/env python
"""Time line of the last N kernel dispatches of a rocprofv3 (rocpd sqlite) kernel trace: start relative to the first, duration, gap to the previous end.

    python tools/rocpd_timeline.py trace.db [N]
"""
import re, sqlite3, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main(path, n=40):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {ncol}, start, end from kernels order by start").fetchall()[-n:]
    t0, prev = rows[0][1], None
    for name, s, e in rows:
        gap = "" if prev is None else f"{(s - prev) / 1e3:8.2f}"
        print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f}  gap {gap:>8s}  {short(name)}")
        prev = max(e, prev) if prev is not None else e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
