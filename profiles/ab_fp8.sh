#!/bin/bash
# A/B of the opt-in FP8 cross terms against the default trunk, same box, alternating: 512 games (4 lanes) and the fill layout
run() { timeout -k 10 300 python bench.py "$@" --no-configs --no-fill --no-games-leg --no-cpu-baseline --timeline 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); rf=r['roofline']; print(r['config']['workload'].split('_')[-1], r['config']['lanes'], 'lanes', r['value'], r['regions_sims_per_sec'], 'frac', rf['frac'], 'launch ms', rf.get('avg_launch_ms'))"; }
for rep in 1 2; do
  for a in split_f16 split_f16_fp8; do
    echo "== $a"; run --net-algo $a || exit 1
    run --net-algo $a --lanes 2 --games 1536 --warmup 3 || exit 1
  done
done
for l in 2 3; do echo "== split_f16_fp8 lanes $l"; run --net-algo split_f16_fp8 --lanes $l || exit 1; done
echo "== split_f16_fp8 256 games (resident)"; run --net-algo split_f16_fp8 --games 256 || exit 1
echo "== split_f16 256 games (resident)"; run --net-algo split_f16 --games 256 || exit 1
