#!/bin/bash
# MFMA utilisation of the trunk from PMC counters (own passes, --pmc only): bash profiles/pmc_mfma.sh r01
# One lane, eager, so every dispatch of k_trunk_split runs alone on the chip.
set -u
ROUND=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2> /dev/null | grep -i -E "MFMA|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_INSTS_VALU\b" | head -60 > "$OUT/pmc_available.txt"
ARGS="--no-cpu-baseline --no-games-leg --no-literal-config --steps 1 --warmup 0 --playouts 40 --graph 0 --lanes 1 --games 512"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_mfma" -o p -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_mfma.err"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pmc_mfma_trace" -o t -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/pmc_mfma.err"
cd "$ROOT" && python3 profiles/summarise_mfma.py "$OUT"
