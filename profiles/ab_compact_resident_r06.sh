#!/bin/bash
# Round 6: small boards.  The resident search on the compact LDS grid, two games per CU, ONE lane (--lanes 1) against the lane table's
# layout of the two-launch step for the same batch (RZ_RESIDENT=0).  M simulations / s.
#   bash profiles/ab_compact_resident_r06.sh > gpurun_out/ab_compact_resident.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0 --warmup 6"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  %.2f ms/move  lanes=%s  %s' % (d['value']/1e6, d['ms_per_step'], d['config'].get('lanes'), (d.get('roofline') or {}).get('kernel','')[:50]))"; }
for G in ${GAMES:-192 256 384 512 768 1024 2048}; do
    echo "== connect4 400 sims, $G games, one resident lane:   $($B --game connect4 --playouts 400 --games $G --steps 12 --lanes 1 2>/dev/null | val)"
    echo "== connect4 400 sims, $G games, the lane table:        $(RZ_RESIDENT=0 $B --game connect4 --playouts 400 --games $G --steps 12 2>/dev/null | val)"
done
for G in 256 512 1024; do
    echo "== 6x6 n4 400 sims, $G games, one resident lane:   $($B --board 6 --playouts 400 --games $G --steps 12 --lanes 1 2>/dev/null | val)"
    echo "== 6x6 n4 400 sims, $G games, the lane table:        $(RZ_RESIDENT=0 $B --board 6 --playouts 400 --games $G --steps 12 2>/dev/null | val)"
done
