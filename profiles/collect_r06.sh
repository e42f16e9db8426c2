#!/bin/bash
# Round 6: collects what is kept under profiles/r06/ on a GPU box (run from the repository root):
#   bash profiles/collect_r06.sh [quick]
# Raw rocprofv3 output goes to /tmp/rz_r06 on the box; profiles/summarise_r06.py picks the files to keep and they come back under
# gpurun_out/r06/ (copy them to profiles/r06/).  rocprofv3 is always given the program itself after `--`; counters get their own
# passes (one counter per pass, no trace domains beside them).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r06
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/r06"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0"

# 1. the default line exactly as the driver runs it
if [ "${1:-}" != quick ]; then
python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "default line done"
fi

# 2. per-kernel times.  default: 512 games on ONE resident lane (k_delta_res: a launch per move, inside the whole-move hipGraph);
#    lanes4: round 5's layout with this round's trunk (four lanes of the two-launch step: k_trunk_delta + k_tree_step_def), eager so
#    that every dispatch is a record; full: the same with the full-board trunk (k_trunk_rows) for the A / B; fill; 256 games
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -o s -- $B --steps 8 --warmup 2 > "$OUT/bench_under_rocprof.json" 2> /dev/null
echo "stats default done"
RZ_RESIDENT=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_lanes4_eager" -o s -- $B --graph 0 --steps 2 > "$OUT/bench_lanes4_eager_under_rocprof.json" 2> /dev/null
RZ_RESIDENT=0 RZ_NET_DELTA=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_trunk_eager" -o s -- $B --graph 0 --steps 2 > "$OUT/bench_full_trunk_eager_under_rocprof.json" 2> /dev/null
echo "stats lanes done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fill" -o s -- $B --games 1536 --steps 6 --warmup 2 > "$OUT/bench_fill_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_256" -o s -- $B --steps 4 --warmup 2 --games 256 > "$OUT/bench_256_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o s -- $B --steps 32 --warmup 8 --board 9 --playouts 200 --games 64 --lanes 1 > "$OUT/bench_c2_under_rocprof.json" 2> /dev/null
echo "kernel stats done"

# 3. HBM traffic counters: separate FETCH_SIZE / WRITE_SIZE passes, eager launches
pmc() {  # tag, bench flags (env before the call)
    tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- $B --graph 0 --steps 1 --warmup 1 "$@" > "$OUT/pmc_${tag}.json" 2> /dev/null
        echo "pmc $tag $c done"   # (a line a minute: a silent run is taken to be hung)
    done
}
if [ "${1:-}" != quick ]; then
pmc default
RZ_RESIDENT=0 pmc lanes4
pmc fill --games 1536
fi
cd "$ROOT" && python3 profiles/summarise_r06.py "$OUT" && cp "$OUT"/keep/* "$ROOT/gpurun_out/r06/"
ls "$ROOT/gpurun_out/r06"
