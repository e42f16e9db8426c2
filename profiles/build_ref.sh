#!/bin/bash
# build the HIP library of another revision (for same-box A / B runs through RZ_HIP_LIBRARY): profiles/build_ref.sh <rev> <out.so>
set -e
rev=$1; out=$2; tmp=$(mktemp -d)
mkdir -p $tmp/csrc $tmp/include
for f in rz_engine.hip rz_net.hip rz_muzero.hip; do git show $rev:rlzero_amd/csrc/$f > $tmp/csrc/$f; done
git show $rev:include/rlzero_hip.h > $tmp/include/rlzero_hip.h
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -shared -Wno-unused-function \
    -I$tmp/include $tmp/csrc/rz_engine.hip $tmp/csrc/rz_net.hip $tmp/csrc/rz_muzero.hip -o $out
rm -rf $tmp
