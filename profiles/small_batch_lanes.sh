#!/bin/bash
run() { timeout -k 10 300 python bench.py "$@" --regions 3 --no-configs --no-fill --no-games-leg --no-cpu-baseline --timeline 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('$*', '->', r['value'], r['regions_sims_per_sec'], 'resident', r['config']['resident_search'])"; }
for l in 1 2 4; do run --board 9 --playouts 200 --games 64 --lanes $l --steps 8 --warmup 8 || exit 1; done
for l in 1 2 4; do run --board 3 --playouts 25 --games 16 --lanes $l --steps 9 --warmup 20 || exit 1; done
for l in 1 2 3 4; do run --games 256 --lanes $l --steps 3 --warmup 2 || exit 1; done
for l in 1 2 4; do run --games 128 --lanes $l --steps 3 --warmup 2 || exit 1; done
