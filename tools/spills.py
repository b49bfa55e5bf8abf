#!/usr/bin/env python
"""Register / scratch report of the built library's code objects (llvm-readelf --notes on the bundled gfx950 images).
    python tools/spills.py [name-substring ...]        -> name, vgprs, spilled vgprs, scratch bytes per lane, LDS"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("NAMP_LIB_PATH") or os.path.join(ROOT, "na_mpnn_amd", "lib", "libnamp_hip.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    import shutil
    objs = []
    # every object of the link carries its own bundle section; llvm-objdump extracts them NEXT TO ITS INPUT: work on a copy in the temp directory
    shutil.copy(so, os.path.join(d, "lib.so"))
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", os.path.join(d, "lib.so")], capture_output=True, text=True, cwd=d)
    for f in sorted(os.listdir(d)):
        if "gfx950" in f:
            objs.append(os.path.join(d, f))
    rows = []
    for o in objs:
        txt = subprocess.run([f"{llvm}/llvm-readelf", "--notes", o], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            rows.append((g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
flt = sys.argv[1:]
for name, v, sp, scr, lds in sorted(set(rows)):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and not any(f in dem for f in flt):
        continue
    print(f"{dem[:110]:110s} vgpr={v:>4s} spill={sp:>4s} scratch={scr:>5s} lds={lds}")
