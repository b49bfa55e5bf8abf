#!/bin/bash
# Build ablation variants of libnamp_hip.so into tools/_variants/ (git-ignored; ships with the gpurun snapshot).
# Only one translation unit (UNIT, default namp.hip) is recompiled per variant; the other objects come from the regular build (na_mpnn_amd/lib/obj).
#   [UNIT=namp_train.hip] tools/build_variants.sh name:-DFLAG[,-DFLAG2] ...        (default: the bf16 storage kernel's ablation set)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_variants
# at most 8 compiles run at once on this container
UNIT=${UNIT:-namp.hip}
OTHERS=""
for u in namp namp_train namp_train_eu namp_persist; do [ "$u.hip" = "$UNIT" ] || OTHERS="$OTHERS na_mpnn_amd/lib/obj/$u.o"; done
build() {
  name=$1; shift
  ( timeout 1500 hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c na_mpnn_amd/csrc/$UNIT -o tools/_variants/$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc tools/_variants/$name.o $OTHERS -o tools/_variants/$name.so &&
    rm -f tools/_variants/$name.o ) &
}
if [ $# -eq 0 ]; then
  set -- nogelu:-DNAMP_ABL_NOGELU nogemm:-DNAMP_ABL_NOGEMM noldsw:-DNAMP_ABL_NOLDSW noprolog:-DNAMP_ABL_NOPROLOG \
         nogemm_nogelu:-DNAMP_ABL_NOGEMM,-DNAMP_ABL_NOGELU skeleton:-DNAMP_ABL_NOGEMM,-DNAMP_ABL_NOGELU,-DNAMP_ABL_NOPROLOG
fi
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  build $name ${flags//,/ }
done
wait
ls -la tools/_variants
