#!/bin/bash
# Lanes of games x hardware queues x batch size on one MI355X (what profiles/r03/lane_sweeps.txt holds and selfplay.plan_lanes is
# built on).  Run from the repository root:   bash profiles/lane_sweep.sh > gpurun_out/lane_sweeps.txt
F="--no-cpu-baseline --no-fill --no-configs --no-games-leg --steps 6 --warmup 3 --regions 1"
run() {  # queues, bench flags
    q=$1; shift
    GPU_MAX_HW_QUEUES=$q python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q $*:', round(d['value']/1e6, 3), d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_ms'))"
}
echo "## 512 games: hardware queues x lanes"
for q in 4 8 16; do for l in 2 3 4 5 6; do run $q --lanes $l; done; done
echo "## batch size x lanes (8 queues; '--trunk-wgs 224' = two capped lanes)"
for G in 128 192 256 320 384 448 512 576 640 704 768 1024 1536; do
    for l in 1 2 3 4; do run 8 --games $G --lanes $l; done
    run 8 --games $G --lanes 2 --trunk-wgs 224
done
echo "## Connect4 (512 games, 400 simulations) and the PUCT rule at the headline geometry"
for l in 2 3 4; do run 8 --game connect4 --playouts 400 --games 512 --lanes $l; run 8 --score-mode puct --lanes $l; done
