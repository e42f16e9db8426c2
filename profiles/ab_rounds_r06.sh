#!/bin/bash
# Round 6: batches beyond two games per CU.  k_delta_res on ONE lane, the grid running in rounds (--lanes 1), against the lane table's
# layout for the same batch (two / four lanes of the two-launch step: k_trunk_delta + k_tree_step_def).  M simulations / s.
#   bash profiles/ab_rounds_r06.sh > gpurun_out/ab_rounds.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0 --warmup 2"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  %.1f ms/move  lanes=%s  %s' % (d['value']/1e6, d['ms_per_step'], d['config'].get('lanes'), (d.get('roofline') or {}).get('kernel','')[:40]))"; }
for G in ${GAMES:-576 640 768 1024 1536 2048 4096}; do
    steps=$(( 4096 / G + 3 ))
    echo "== $G games, one resident lane:   $($B --games $G --steps $steps --lanes 1 2>/dev/null | val)"
    echo "== $G games, the lane table (r05): $(RZ_RESIDENT=0 $B --games $G --steps $steps 2>/dev/null | val)"
done
