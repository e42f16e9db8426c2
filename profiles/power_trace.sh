#!/bin/bash
# Power / clock telemetry while the bench runs (read-only rocm-smi polling): bash profiles/power_trace.sh r03 [bench flags]
# e.g. the headline (four lanes of 128 games), `--games 1536` (the fill), `--lanes 2`
ROUND=${1:-r03}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$ROUND
TAG=$(echo "$*" | tr -c 'a-zA-Z0-9' '_' | sed 's/_*$//'); TAG=${TAG:-default}
mkdir -p "$OUT"
(rocm-smi --showpower --showclocks --showmaxpower --showtemp 2>&1 | grep -vE "^=|^$" | head -40) > "$OUT/power_idle.txt"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-fill --no-configs --no-games-leg --regions 1 --steps ${STEPS:-300} --warmup 2 "$@" > "$OUT/power_bench_$TAG.json" 2> /dev/null &
BPID=$!
sleep 10
: > "$OUT/power_trace_$TAG.txt"
for i in $(seq 1 24); do
    if ! kill -0 $BPID 2> /dev/null; then break; fi
    (rocm-smi --showpower --showclocks 2>&1 | grep -E -i "power|sclk" | sed 's/^GPU\[0\][[:space:]]*: //' | tr '\n' ';'; echo) >> "$OUT/power_trace_$TAG.txt"
    sleep 0.3
done
wait $BPID
python3 -c "import json;d=json.loads(open('$OUT/power_bench_$TAG.json').read().strip().splitlines()[-1]);print('$TAG', round(d['value']/1e6,3),'M sims/s', d['ms_per_step'],'ms per move')" >> "$OUT/power_trace_$TAG.txt"
tail -4 "$OUT/power_trace_$TAG.txt"
