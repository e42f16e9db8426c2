#!/bin/bash
# Power / clock telemetry while the default bench runs (read-only rocm-smi polling): bash profiles/power_trace.sh r01
ROUND=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
(rocm-smi --showpower --showclocks --showmaxpower --showtemp 2>&1 | head -60) > "$OUT/power_idle.txt"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-literal-config --no-games-leg --steps 160 --warmup 2 > "$OUT/power_bench.json" 2> /dev/null &
BPID=$!
sleep 9
: > "$OUT/power_trace.txt"
for i in $(seq 1 30); do
    if ! kill -0 $BPID 2> /dev/null; then break; fi
    (date +%s.%N; rocm-smi --showpower --showclocks 2>&1 | grep -E -i "power|sclk|mclk|fclk" | head -8) >> "$OUT/power_trace.txt"
    sleep 0.3
done
wait $BPID
cat "$OUT/power_bench.json" | cut -c1-120
