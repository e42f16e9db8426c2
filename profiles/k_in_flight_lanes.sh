#!/bin/bash
run() { timeout -k 10 300 python bench.py "$@" --regions 2 --no-configs --no-fill --no-games-leg --no-cpu-baseline --timeline 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('$*', '->', round(r['value']/1e6,3), [round(x/1e6,2) for x in r['regions_sims_per_sec']], 'lanes', r['config']['lanes'])"; }
for g in 32 128; do for l in 1 2 3 4; do run --board 9 --playouts 200 --games $g --steps 8 --warmup 8 --in-flight 16 --lanes $l || exit 1; done; done
for l in 2 3 4; do run --board 9 --playouts 200 --games 64 --steps 8 --warmup 8 --in-flight 16 --lanes $l || exit 1; done
for l in 1 2 3 4; do run --board 9 --playouts 200 --games 64 --steps 8 --warmup 8 --in-flight 8 --lanes $l || exit 1; done
