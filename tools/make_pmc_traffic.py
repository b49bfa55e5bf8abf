#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, as the PMC slots require).

    python tools/make_pmc_traffic.py <fetch.db> <write.db> <commit> <round> [<cfg5 fetch.db> <cfg5 write.db> <precision>]... > profiles/pmc_traffic.json

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


EU_SLICES = int(__import__("os").environ.get("NAMP_TRAIN_EU_SLICES", "2"))
KIND = {(0, 1): "enc_message", (0, 3): "enc_edge_message", (1, 3): "enc_edge_dec_message", (1, 0): "dec_message"}


# cfg5 (B = 16, N = 1500, K = 48): compulsory HBM bytes per launch of the per-edge backward kernels — rows that MUST move (h_E rows in, the
# other consumer's dL/dh_E rows in where added, dL/dh_E rows out, G1 rows out for the table-gradient gather, E_idx, per-tile sums; the
# residue tables count as cache-resident like in SURVEY 8(d)).  The round-3 kernels additionally write A1 / G2 (/ A2 / G3) rows that exist
# only to be re-read by the row contractions: NOT compulsory, so their measured / algorithmic ratio shows the waste.
def cfg5_algorithmic(kernel, prec):
    E, G = 16 * 1500 * 48, 16 * 1500
    row16 = 256 if prec == "bf16" else 512            # G1 row bytes
    base = E * 4 + E // 16 * 512 + 3 * G * 512
    if "edge_update_bwd_a16" in kernel:                # launch A: h_E + dL/dh_E' rows in, dL/dx rows (fp32) + G2 rows (bf16) out, E_idx, tables
        return E * (512 + 512 + 512 + 256) + E * 4 + 3 * G * 512
    if "edge_update_bwd_b16" in kernel:                # launch B: h_E + G2 + dL/dx rows in, dL/dh_E + G1 rows out, E_idx, per-tile sums, tables
        return E * (512 + 256 + 512 + 512 + 256) + base
    if "feat_wgrad" in kernel:                         # g_pre rows once (bf16 operand tiles in the mixed-precision mode), atom frames, E_idx, the 128 chunks' [128 x 5200] partials
        return E * (256 if prec == "bf16" else 512) + G * 18 * 16 + E * 4 + 128 * 128 * 5200 * 4
    if "embed_ln_bwd" in kernel:                       # g and y rows in, g_pre rows out (+ its bf16 tiles in the mixed-precision mode), row statistics
        return E * (512 * 3 + (256 if prec == "bf16" else 0) + 8)
    if "pos_grad_kernel" in kernel:
        return E * 512 + E * 4
    if "scatter_rows_kernel" in kernel:                # G1 rows once, reverse adjacency, one or two [G,128] outputs
        return E * row16 + E * 4 + 2 * G * 512
    if "edge_bwd_dw" in kernel:
        acc = "true" in kernel                         # <MODE, [PREC,] ACC, GPA>
        return E * (512 + (512 if acc else 0) + row16 + 512) + base
    if "edge_chain_bwd_kernel<3" in kernel:            # edge update: h_E in, dL/dh_E' in, dL/dh_E out, G1 out
        return E * (512 + 512 + 512 + row16) + base
    if "edge_chain_bwd_kernel<0" in kernel or "edge_chain_bwd_kernel<1" in kernel:
        return E * (512 + 512 + 512 + row16) + base
    return None


def cfg5_section(fetch_db, write_db, prec):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    sec = {}
    for k in sorted(set(f) | set(w)):
        if not any(t in k for t in ("edge_bwd_dw", "edge_update_bwd_", "edge_chain_bwd", "wgrad", "scatter_rows", "edge_mlp_x3_persistent",
                                    "edge_mlp_bf16_persistent", "feat_wgrad", "edge_features", "reduce_sum", "pos_grad", "ln_rows", "embed_ln_bwd")):
            continue
        nbytes = (2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024
        e = {"measured_bytes": round(nbytes)}
        if "edge_chain_bwd_kernel<3" in k and prec == "x3":
            # split-bf16: train.EDGE_UPDATE_SLICES (default 2) launches walk the batch; bytes per STAGE = the per-launch average x slices
            nbytes *= EU_SLICES
            e = {"measured_bytes": round(nbytes), "launches_per_stage": EU_SLICES}
        alg = cfg5_algorithmic(k, prec)
        if alg:
            e["algorithmic_bytes"] = alg
            e["measured_over_algorithmic"] = round(nbytes / alg, 3)
        sec[k.replace("void ", "")[:70]] = e
    return sec


def main(fetch_db, write_db, commit, rnd, *extra):
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
    for i in range(0, len(extra) - 2, 3):
        out["cfg5_" + extra[i + 2]] = cfg5_section(extra[i], extra[i + 1], extra[i + 2])
    if extra:
        out["_cfg5_source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --workload cfg5 --precision <p> --steps 2 "
                               "--warmup 1`; per-dispatch averages per kernel; algorithmic = compulsory rows (tools/make_pmc_traffic.py:cfg5_algorithmic)")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
