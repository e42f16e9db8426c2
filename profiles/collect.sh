#!/bin/bash
# Collects the measurements kept under profiles/<round>/ on a GPU box (run from the repository root):
#   bash profiles/collect.sh r01
# Output goes to gpurun_out/<round>/ (scratch); copy what is to be kept into profiles/<round>/.
# rocprofv3 is always given the program itself after `--` and counters get their own passes.
set -u
ROUND=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

# 1. the default bench line (with the CPU baseline) and the literal 4096/8 games-per-GPU configuration
python3 "$ROOT/bench.py" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --lanes 1 --games 512 > "$OUT/bench_1lane_512games.json" 2>> "$OUT/bench_default.err"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --games 896 > "$OUT/bench_2lanes_896games.json" 2>> "$OUT/bench_default.err"
# the f32-input MFMA trunk (Winograd F(4x4,3x3)) on the geometry it was tuned for, for comparison
python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --net-algo winograd_f4 --games 896 > "$OUT/bench_f32_winograd_f4.json" 2>> "$OUT/bench_default.err"

# 2. per-kernel times of the same default command (hipGraph replays: rocprofv3 attributes the time a
#    dependent node waits to the node) and of an eager run (true durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -o s -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config > "$OUT/bench_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager" -o s -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --graph 0 --steps 2 > "$OUT/bench_eager_under_rocprof.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_eager_1lane" -o s -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --graph 0 --steps 2 --lanes 1 --games 512 \
    > "$OUT/bench_eager_1lane_under_rocprof.json" 2> /dev/null

# 3. HBM traffic counters, one pass each (short eager run of the default geometry)
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- \
        python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --steps 1 --warmup 0 --playouts 40 --graph 0 \
        > /dev/null 2> /dev/null
done

# 3b. sweeps behind the defaults (games per GPU / trunk workgroups; heads GEMM variants x lane layouts)
(cd "$ROOT" && bash profiles/sweep_games.sh > "$OUT/sweep_games.txt" 2>&1; bash profiles/sweep_heads.sh > "$OUT/sweep_heads.txt" 2>&1)

# 4. the other single-GPU configurations of BASELINE.json (parity-test cases, recorded for reference)
python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --lanes 1 --board 3 --playouts 25 --games 1 > "$OUT/bench_c1_ttt.json" 2> /dev/null
python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --lanes 1 --board 9 --playouts 200 --games 64 > "$OUT/bench_c2_9x9.json" 2> /dev/null
python3 "$ROOT/bench.py" --no-cpu-baseline --no-games-leg --no-literal-config --lanes 1 --game connect4 --playouts 400 --games 512 > "$OUT/bench_c3_connect4.json" 2> /dev/null
python3 "$ROOT/bench.py" --game muzero --steps 8 --warmup 2 > "$OUT/bench_c5_muzero_cartpole.json" 2> /dev/null

# 5. the micro-benchmarks behind the trunk's design (profiles/microbench/README.md)
if [ -z "${SKIP_MICRO:-}" ]; then
    for m in f32_mfma_overlap f16_mfma_rate f16_mfma_fillers; do
        hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_$m "$ROOT/profiles/microbench/$m.hip" 2> /dev/null && \
            timeout 200 /tmp/mb_$m > "$OUT/microbench_$m.txt" 2>&1
    done
fi

cd "$ROOT" && python3 profiles/summarise.py "$OUT"

# 6. the default line once more, now that this run's counters are in place (bench.py reports roofline.traffic from
#    profiles/<round>/pmc_traffic.json)
if [ -f "$OUT/keep/pmc_traffic.json" ]; then
    cp "$OUT/keep/pmc_traffic.json" "$ROOT/profiles/r01/pmc_traffic.json"
    python3 "$ROOT/bench.py" > "$OUT/keep/bench_default.json" 2>> "$OUT/bench_default.err"
fi
