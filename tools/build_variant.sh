#!/bin/bash
# usage: tools/build_variant.sh <tag> <file.hip> [extra hipcc flags]   -> variants/libsga_<tag>.so
# Experiment helper: rebuild ONE source with extra flags / -D switches and link it with the other objects.
# Run with SGA_LIB_PATH=variants/libsga_<tag>.so to load it instead of csrc/libsga_hip.so.
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p variants
C=sgaligner_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -I $C "$@" -c $C/$src -o variants/${src%.hip}_$tag.o 2>&1 | grep -E "error" || true
objs=""
for f in $C/*.hip; do b=$(basename $f .hip); if [ "$b.hip" == "$src" ]; then objs="$objs variants/${b}_$tag.o"; else objs="$objs $C/$b.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsga_$tag.so $objs
echo built variants/libsga_$tag.so
