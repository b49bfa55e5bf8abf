#!/bin/bash
# Build ablation variants of libnamp_hip.so into gpurun_out/variants/ (scratch; ships with the gpurun snapshot
# only if placed elsewhere, so variants are written to tools/_variants/ which is git-ignored).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_variants
build() { name=$1; shift; hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" na_mpnn_amd/csrc/namp.hip -o tools/_variants/$name.so & }
build base
build nogelu -DNAMP_ABL_NOGELU
build layers1 -DNAMP_ABL_LAYERS=1
build layers2 -DNAMP_ABL_LAYERS=2
build noprolog -DNAMP_ABL_NOPROLOG
build noprolog_nogelu -DNAMP_ABL_NOPROLOG -DNAMP_ABL_NOGELU
build w1l2 -DNAMP_ABL_W1_FROM_L2
wait
ls -la tools/_variants
