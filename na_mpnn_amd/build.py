"""In-tree build of libnamp_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m na_mpnn_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libnamp_hip.so")
INC = os.path.join("..", "..", "include", "namp.h")
# translation unit -> files it depends on (each unit is compiled to its own object, in parallel, then linked)
UNITS = {"namp.hip": ["namp.hip", "namp_kernels.h", "namp_bf16s32.h", "namp_bf16p.h", "namp_node_w.h", "namp_order.h", "namp_device.h", INC],
         "namp_persist.hip": ["namp_persist.hip", "namp_kernels.h", "namp_device.h"],
         "namp_train.hip": ["namp_train.hip", "namp_train.h", "namp_train_dw.h", "namp_device.h", INC],
         "namp_train_eu.hip": ["namp_train_eu.hip", "namp_train_eu.h", "namp_train_dw.h", "namp_train.h", "namp_device.h", INC]}
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
OBJ_DIR = os.path.join(HERE, "lib", "obj")
TIMEOUT_S = 1500


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(os.path.join(CSRC, d)) < t for d in deps)     # equal mtimes: rebuild (mtime granularity)


def _obj(unit):
    return os.path.join(OBJ_DIR, unit.replace(".hip", ".o"))


def up_to_date():
    return _newer(OUT, sorted({d for deps in UNITS.values() for d in deps}))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()
    procs = []
    for unit, deps in UNITS.items():
        if not force and _newer(_obj(unit), deps):
            continue
        cmd = [cc] + CFLAGS + ["-c", os.path.join(CSRC, unit), "-o", _obj(unit)]
        if verbose:
            print("[na_mpnn_amd.build]", " ".join(cmd), flush=True)
        procs.append((unit, subprocess.Popen(cmd, cwd=CSRC)))
    try:
        for unit, p in procs:
            try:
                rc = p.wait(timeout=TIMEOUT_S)
            except subprocess.TimeoutExpired:
                raise RuntimeError(f"hipcc timed out on {unit} after {TIMEOUT_S}s")
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {unit} (exit {rc})")
    finally:
        # never leave a compiler running behind a failure: it would overwrite lib/obj/*.o under a retry
        for _, p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + [_obj(u) for u in UNITS] + ["-o", OUT]
    if verbose:
        print("[na_mpnn_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC, timeout=TIMEOUT_S)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
