set -e
rm -rf /tmp/tp && mkdir -p /tmp/tp && cp -r rlzero_amd/csrc include /tmp/tp/ && cd /tmp/tp && python3 /root/repo/profiles/microbench/tree_ticks/patch_tp.py
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -Wno-unused-function"
for f in rz_engine rz_net rz_muzero; do hipcc $FLAGS -DRZ_NET_PROFILE -DRZ_SOURCE_HASH='"tp"' -Iinclude -c csrc/$f.hip -o $f.o 2> $f.err & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC rz_engine.o rz_net.o rz_muzero.o -o /root/repo/scratch/librz_tp.so
