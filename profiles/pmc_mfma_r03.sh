#!/bin/bash
# MFMA utilisation of k_trunk_rows from PMC counters (own pass, --pmc only) + the kernel trace of the same command:
#   bash profiles/pmc_mfma_r03.sh        (one lane of 768 games, eager: every trunk dispatch runs alone on the chip, 3 boards per workgroup)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r03_mfma
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/r03"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --steps 1 --warmup 0 --playouts 40 --graph 0 --lanes 1 --games 768"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_mfma" -o p -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_mfma.err"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_mfma_trace" -o t -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/pmc_mfma.err"
cd "$ROOT" && python3 profiles/summarise_mfma.py "$OUT" k_trunk_rows 768 && cp "$OUT/keep/pmc_mfma.json" "$ROOT/gpurun_out/r03/pmc_mfma.json"
