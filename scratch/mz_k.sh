#!/bin/bash
run() { timeout -k 10 300 python bench.py --game muzero --playouts 50 --games 8192 --steps 2048 --warmup 256 --no-cpu-baseline --no-configs --no-fill --no-games-leg "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); rf=r['roofline']; print('$*', '->', round(r['value']/1e9,4), 'G; launch ms', rf.get('avg_launch_ms'), 'span', r.get('gpu_span_ms'), 'wall', r.get('region_wall_ms'))"; }
for k in 16 32 64 16 32 64; do run --mz-moves-per-launch $k || exit 1; done
