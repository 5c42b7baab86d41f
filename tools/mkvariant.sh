#!/bin/bash
# A library variant for A/B runs on one box: kernels.hip compiled with extra flags, linked with the tree's other objects.
#   bash tools/mkvariant.sh <name> "<extra hipcc flags>"   ->  variants/<name>.so   (then tools/variants2.sh <name> ... on the GPU box)
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$here/readsb_amd/csrc" -j8 || exit 1
mkdir -p "$here/variants" "$here/build/obj"
for src in kernels.hip capi.cpp; do  # (capi.cpp prints what the RSB_STAGE_CLOCKS kernels count)
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950 $* -c "$here/readsb_amd/csrc/$src" -o "$here/build/obj/var_${name}_$src.o" 2>&1 | grep -E "error|spill" &
done
wait
objs=$(ls "$here"/build/obj/*.o | grep -v "/var_\|kernels.hip.o\|capi.cpp.o")
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o "$here/variants/$name.so" "$here/build/obj/var_${name}_kernels.hip.o" "$here/build/obj/var_${name}_capi.cpp.o" $objs && echo "variants/$name.so"
