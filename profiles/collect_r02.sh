#!/bin/bash
# Round 2: collects what is kept under profiles/r02/ on a GPU box (run from the repository root):
#   bash profiles/collect_r02.sh
# Output goes to gpurun_out/r02/ (scratch); profiles/summarise_r02.py picks the files to keep.
# rocprofv3 is always given the program itself after `--` and counters get their own passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"

# 1. the default line exactly as the driver runs it (CPU baseline, literal configs[3] share, C1 / C2 / C3 / C5 legs)
python3 "$ROOT/bench.py" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "default line done"

# 2. per-kernel times: the default command under rocprofv3 (hipGraph replays) and eager runs (true durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -o s -- $B > "$OUT/bench_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager" -o s -- $B --graph 0 --steps 2 > "$OUT/bench_eager_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager_1lane" -o s -- $B --graph 0 --steps 2 --lanes 1 --games 512 > "$OUT/bench_eager_1lane_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager_literal" -o s -- $B --graph 0 --steps 2 --lanes 2 --games 512 > "$OUT/bench_eager_literal_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager_c2_k16" -o s -- $B --graph 0 --steps 4 --board 9 --playouts 200 --games 64 --lanes 1 --in-flight 16 > "$OUT/bench_eager_c2_k16_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_muzero" -o s -- $B --game muzero --games 8192 --steps 64 --warmup 16 > "$OUT/bench_muzero_under_rocprof.json" 2> /dev/null
echo "kernel stats done"

# 3. HBM traffic counters, one pass each (short eager run of the default geometry)
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- $B --steps 1 --warmup 0 --playouts 40 --graph 0 > /dev/null 2> /dev/null
done
echo "pmc done"

# 4. lane layouts of the literal 512 games per GPU and of large batches; in-flight sweep of configs[1]
{
  for spec in "--lanes 1 --games 512" "--lanes 2 --games 512 --trunk-wgs 224" "--lanes 2 --games 512 --heads-algo split32" \
              "--lanes 2 --games 512" "--lanes 3 --games 513 --heads-algo parts" \
              "--lanes 2 --games 1344 --trunk-wgs 224" "--lanes 2 --games 1024" "--lanes 2 --games 1536"; do
    echo "== $spec"; $B $spec 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r.get('small_kernels'))"
  done
} > "$OUT/lane_sweeps.txt" 2>&1
{
  for K in 1 2 4 8 16; do
    echo "== configs[1] 9x9 / 200 sims / 64 games, $K in flight"; $B --board 9 --playouts 200 --games 64 --lanes 1 --steps 8 --warmup 2 --in-flight $K 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])"
  done
} > "$OUT/in_flight_sweep.txt" 2>&1
echo "sweeps done"
cd "$ROOT" && python3 profiles/summarise_r02.py "$OUT"
