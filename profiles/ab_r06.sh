#!/bin/bash
# A / B of library variants on ONE box, alternating: bash profiles/ab_r06.sh <variant.so> [<variant.so> ...]   (the first entry "base" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 3 --timeline 0 --steps 8 --warmup 3"
for rep in 1 2; do
  for lib in base "$@"; do
    if [ "$lib" = base ]; then unset RZ_HIP_LIBRARY; else export RZ_HIP_LIBRARY="$ROOT/$lib"; fi
    $B ${AB_FLAGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-44s %.3f M sims/s  launch %.3f ms  %.2f GHz  %s W' % ('$lib', d['value']/1e6, r.get('avg_launch_ms',0), r.get('sclk_in_loop_ghz',0) or 0, r.get('board_power_w')))"
  done
done
