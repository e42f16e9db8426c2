#!/bin/bash
run() { timeout -k 10 300 python bench.py "$@" --regions 2 --no-configs --no-fill --no-games-leg --no-cpu-baseline --timeline 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('$*', '->', round(r['value']/1e6,3), [round(x/1e6,2) for x in r['regions_sims_per_sec']], 'lanes', r['config']['lanes'], r['config']['launches_per_step'])"; }
for l in 2 4 2 4; do run --game connect4 --playouts 400 --games 512 --steps 6 --warmup 6 --lanes $l || exit 1; done
for g in 384 768 1024; do for l in 2 3 4; do run --game connect4 --playouts 400 --games $g --steps 4 --warmup 4 --lanes $l || exit 1; done; done
for g in 512 1024; do for l in 2 3 4; do run --board 9 --playouts 200 --games $g --steps 6 --warmup 6 --lanes $l || exit 1; done; done
