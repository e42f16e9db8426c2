#!/bin/bash
# Instruction counters of the default line's kernels (round 5 kept the same table for k_tree_step_def: tree_step_instruction_counters.txt):
# one counter per rocprofv3 pass, sums over the launch, per launch.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r06i
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0 --graph 0 --steps 1 --warmup 1"
echo "# 512 games x 800 simulations per launch of k_delta_res (409 600 simulations; 2048 waves): per launch"
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -o p -- $B > /dev/null 2> "$OUT/$c.err"
    f=$(find "$OUT/$c" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
        python3 - "$f" "$c" <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    k = 'k_delta_res' if 'k_delta_res' in k else None
    if k and r['Counter_Name'] == sys.argv[2]:
        tot[k] += float(r['Counter_Value']); n[k] += 1
for k in tot: print('%-22s %-12s launches %3d  per launch %.5g  per simulation of a game %.1f' % (sys.argv[2], k, n[k], tot[k] / n[k], tot[k] / n[k] / 409600.0))
PY
    else
        echo "$c: no output ($(tail -1 $OUT/$c.err | cut -c1-120))"
    fi
done
