#!/bin/bash
# libsga_hip.so with the sweep3 phase counters (-DS3_DBG_TIMING) -> tools/dbg/libsga_timing.so (use with SGA_LIB_PATH)
set -e
cd "$(dirname "$0")/../../sgaligner_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -I . -DS3_DBG_TIMING "$@" -c sweep3.hip -o /tmp/sweep3_timing.o
objs=$(ls *.o | grep -v '^sweep3.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libsga_timing.so $objs /tmp/sweep3_timing.o
echo built tools/dbg/libsga_timing.so
