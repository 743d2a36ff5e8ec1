#!/bin/bash
# libsga_hip.so with sweep3.hip compiled with extra flags -> tools/dbg/libsga_<tag>.so (use with SGA_LIB_PATH):  build_variant_lib.sh <tag> [flags...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../../sgaligner_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -I . "$@" -c sweep3.hip -o /tmp/sweep3_$tag.o
objs=$(ls *.o | grep -v '^sweep3.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libsga_$tag.so $objs /tmp/sweep3_$tag.o
echo built tools/dbg/libsga_$tag.so
