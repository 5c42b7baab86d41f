#!/bin/bash
# Stage clocks of k_scan and k_replay_emit (ticks / 16 summed over wavefronts) from a library built with -DRSB_STAGE_CLOCKS=1.
# Build it here first (the GPU box only runs it):
#   mkdir -p variants && cd readsb_amd/csrc && /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -Wno-unused-parameter --offload-arch=gfx950 \
#     -DRSB_STAGE_CLOCKS=1 -shared -o ../../variants/clocks.so kernels.hip noalloc.hip capi.cpp modes_host.cpp wire.cpp stream.cpp dropin.cpp
# then: gpurun -- 'bash tools/emit_clocks.sh'    (variants/ is not tracked; delete it afterwards)
cp readsb_amd/libreadsb_amd.so /tmp/keep.so
cp variants/clocks.so readsb_amd/libreadsb_amd.so || exit 1
timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --steps-in-flight 1 2>&1 | grep -E "emit clocks|stage clocks" | tail -3
cp /tmp/keep.so readsb_amd/libreadsb_amd.so
