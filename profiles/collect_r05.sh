#!/bin/bash
# Round 5: collects what is kept under profiles/r05/ on a GPU box (run from the repository root):
#   bash profiles/collect_r05.sh [quick]
# Raw rocprofv3 output goes to /tmp/rz_r05 on the box; profiles/summarise_r03.py picks the files to keep and they come back
# under gpurun_out/r05/ (copy them to profiles/r05/).  rocprofv3 is always given the program itself after `--`; counters get their
# own passes (one counter per pass, no trace domains beside them).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r05
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/r05"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0"

# 1. the default line exactly as the driver runs it
if [ "${1:-}" != quick ]; then
python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "default line done"
fi

# 2. per-kernel times: the headline command under rocprofv3 (hipGraph replays), then eager runs (one dispatch per kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -o s -- $B --steps 6 --warmup 2 > "$OUT/bench_under_rocprof.json" 2> /dev/null
echo "stats default done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fill" -o s -- $B --games 1536 --steps 6 --warmup 2 > "$OUT/bench_fill_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager" -o s -- $B --graph 0 --steps 2 > "$OUT/bench_eager_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager_host_moves" -o s -- $B --graph 0 --steps 2 --device-moves 0 > "$OUT/bench_eager_host_moves_under_rocprof.json" 2> /dev/null
echo "stats eager done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c1" -o s -- $B --steps 180 --warmup 20 --board 3 --playouts 25 --games 1 --lanes 1 > "$OUT/bench_c1_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c1x16" -o s -- $B --steps 180 --warmup 20 --board 3 --playouts 25 --games 16 --lanes 1 > "$OUT/bench_c1x16_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o s -- $B --steps 32 --warmup 8 --board 9 --playouts 200 --games 64 --lanes 1 > "$OUT/bench_c2_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_256" -o s -- $B --steps 4 --warmup 2 --games 256 > "$OUT/bench_256_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager_c3" -o s -- $B --graph 0 --steps 2 --game connect4 --playouts 400 --games 512 > "$OUT/bench_eager_c3_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_muzero" -o s -- $B --game muzero --playouts 50 --games 8192 --steps 256 --warmup 64 > "$OUT/bench_muzero_under_rocprof.json" 2> /dev/null
echo "kernel stats done"

# 3. HBM traffic counters: separate FETCH_SIZE / WRITE_SIZE passes of every workload the line reports, at ITS playout count
pmc() {  # tag, bench flags
    tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- $B --graph 0 --steps 1 --warmup 1 "$@" > "$OUT/pmc_${tag}.json" 2> /dev/null
        echo "pmc $tag $c done"   # (a line a minute: a silent run is taken to be hung)
    done
}
if [ "${1:-}" != quick ]; then
pmc default
pmc 3launch --deferred 0
pmc fill --games 1536
pmc puct --score-mode puct
pmc c2 --board 9 --playouts 200 --games 64 --lanes 1
pmc c2k16 --board 9 --playouts 200 --games 64 --in-flight 16
pmc c3 --game connect4 --playouts 400 --games 512
pmc c1 --board 3 --playouts 25 --games 1 --lanes 1 --steps 8
pmc c1x16 --board 3 --playouts 25 --games 16 --lanes 1 --steps 8
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_c5_$c" -o p -- $B --game muzero --playouts 50 --games 8192 --steps 64 --warmup 32 > "$OUT/pmc_c5.json" 2> /dev/null
done
echo "pmc done"
fi
cd "$ROOT" && python3 profiles/summarise_r03.py "$OUT" && cp "$OUT"/keep/* "$ROOT/gpurun_out/r05/"
# 4. board power and clocks while the final binary runs the headline and the fill (read-only rocm-smi polling)
STEPS=250 bash "$ROOT/profiles/power_trace.sh" r05
STEPS=90 bash "$ROOT/profiles/power_trace.sh" r05 --games 1536
ls "$ROOT/gpurun_out/r05"
