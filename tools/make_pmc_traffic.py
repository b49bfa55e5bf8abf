#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, as the PMC slots require).

    python tools/make_pmc_traffic.py <fetch.db> <write.db> <commit> <round> > profiles/pmc_traffic.json

HBM-side bytes per launch = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads,
MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024, averaged over the dispatches of a kernel.  Kernel -> launch kind of bench.py:
edge_mlp_kernel<MODE, TAIL, PREC, PRE> with PREC 0 = exact fp32 ("cfg2_fp32"), 2 = split-bf16 ("cfg2").
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection").fetchall()
    per = defaultdict(float)
    for k, d, c, v in rows:
        if c == counter:
            per[(re.sub(r"\(.*", "", k), d)] += v
    agg = defaultdict(list)
    for (k, d), v in per.items():
        agg[k].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}


KIND = {(0, 1): "enc_message", (0, 3): "enc_edge_message", (1, 3): "enc_edge_dec_message", (1, 0): "dec_message"}


def main(fetch_db, write_db, commit, rnd):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {"_commit": commit, "_round": rnd,
           "_source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of "
                      "`bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary`; bytes = 2 x FETCH_SIZE KB + WRITE_SIZE KB "
                      "(FETCH_SIZE doubled per MI355X_MICROARCH.md)",
           "cfg2": {}, "cfg2_fp32": {}, "raw_KB": {}}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        out["raw_KB"][k[:80]] = {"FETCH_SIZE": round(fk, 1), "WRITE_SIZE": round(wk, 1)}
        m = re.search(r"edge_mlp_kernel<(\d+), (\d+), (\d+), (\d+)>", k)
        if m:
            mode, tail, prec, pre = map(int, m.groups())
            kind = KIND.get((mode, pre))
            if kind and tail:
                out["cfg2" if prec == 2 else "cfg2_fp32"][kind] = round((2 * fk + wk) * 1024, 1)
        if "gather_cat_kernel" in k:
            out["gather_cfg3"] = round((2 * fk + wk) * 1024, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
