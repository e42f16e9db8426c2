#!/bin/bash
# the library with -DRZ_NET_PROFILE (phase ticks in the trunk kernels) -> profiles/microbench/librlzero_netprof.so
set -e
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -Wno-unused-function -Iinclude"
H=$(python -c "import rlzero_amd._build as b; print(b.source_hash())")
mkdir -p /tmp/netprof
for f in rz_engine rz_net rz_muzero; do
  hipcc $F -DRZ_NET_PROFILE -DRZ_SOURCE_HASH="\"$H\"" -c rlzero_amd/csrc/$f.hip -o /tmp/netprof/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/netprof/*.o -o profiles/microbench/librlzero_netprof.so
