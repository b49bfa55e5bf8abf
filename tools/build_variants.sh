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
build nogelu -DNAMP_ABL_NOGELU
build x1 -DNAMP_ABL_X1
build nogemm -DNAMP_ABL_NOGEMM
build nogemm_nogelu -DNAMP_ABL_NOGEMM -DNAMP_ABL_NOGELU
build nodma -DNAMP_ABL_NODMA
build notail -DNAMP_ABL_NOTAIL
build noprolog -DNAMP_ABL_NOPROLOG
build nostore -DNAMP_ABL_NOSTORE
build noln -DNAMP_ABL_NOLN
build notable2 -DNAMP_ABL_NOTABLE2
build skel -DNAMP_ABL_NOGEMM -DNAMP_ABL_NOGELU -DNAMP_ABL_NOTAIL
build skel_nostore -DNAMP_ABL_NOGEMM -DNAMP_ABL_NOGELU -DNAMP_ABL_NOTAIL -DNAMP_ABL_NOSTORE
build skel_nostore_noprolog -DNAMP_ABL_NOGEMM -DNAMP_ABL_NOGELU -DNAMP_ABL_NOTAIL -DNAMP_ABL_NOSTORE -DNAMP_ABL_NOPROLOG -DNAMP_ABL_NOTABLE2
build skel_min -DNAMP_ABL_NOGEMM -DNAMP_ABL_NOGELU -DNAMP_ABL_NOTAIL -DNAMP_ABL_NOSTORE -DNAMP_ABL_NOPROLOG -DNAMP_ABL_NOTABLE2 -DNAMP_ABL_NOLN -DNAMP_ABL_NODMA
wait
ls -la tools/_variants
