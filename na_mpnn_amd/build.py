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
SOURCES = ["namp.hip"]
DEPS = ["namp.hip", "namp_kernels.h", "namp_device.h", os.path.join("..", "..", "include", "namp.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print("[na_mpnn_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
