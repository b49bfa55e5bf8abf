#!/bin/bash
# Build ablation variants of libnamp_hip.so into tools/_variants/ (git-ignored; ships with the gpurun snapshot).
# Only namp.hip is recompiled per variant; the training objects come from the regular build (na_mpnn_amd/lib/obj).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_variants
# at most 8 compiles run at once on this container
build() {
  name=$1; shift
  ( timeout 1500 hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c na_mpnn_amd/csrc/namp.hip -o tools/_variants/$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc tools/_variants/$name.o na_mpnn_amd/lib/obj/namp_train.o -o tools/_variants/$name.so &&
    rm -f tools/_variants/$name.o ) &
}
build gscalar -DNAMP_GELU_SCALAR
build noslp -fno-slp-vectorize
build gscalar_noslp -DNAMP_GELU_SCALAR -fno-slp-vectorize
wait
ls -la tools/_variants
