#!/bin/bash
run() { timeout -k 10 300 python bench.py "$@" --regions 3 --no-configs --no-fill --no-games-leg --no-cpu-baseline --timeline 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('$*', '->', r['value'], r['regions_sims_per_sec'])"; }
run --board 3 --playouts 25 --games 1 --lanes 1 --steps 9 --warmup 20 || exit 1
run --board 3 --playouts 25 --games 16 --lanes 1 --steps 9 --warmup 20 || exit 1
run --board 9 --playouts 200 --games 64 --lanes 1 --steps 8 --warmup 8 || exit 1
run --games 256 --steps 3 --warmup 2 || exit 1
run --game connect4 --playouts 400 --games 512 --steps 6 --warmup 6 || exit 1
run || exit 1
