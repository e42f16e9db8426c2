#!/bin/bash
# LDS bank-conflict share of the default line's kernels: one counter per pass (no trace domains beside them)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r06l
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/r06"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0 --graph 0 --steps 1 --warmup 1"
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -o p -- $B > /dev/null 2> "$OUT/$c.err"
    f=$(find "$OUT/$c" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
        python3 - "$f" "$c" <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    k = 'k_delta_res' if 'k_delta_res' in k else 'k_trunk_delta' if 'k_trunk_delta' in k else 'k_heads_split' if 'k_heads_split' in k else 'k_deferred_priors' if 'k_deferred_priors' in k else None
    if k and r['Counter_Name'] == sys.argv[2]:
        tot[k] += float(r['Counter_Value']); n[k] += 1
for k in tot: print('%-28s %-18s launches %3d  per launch %.4g' % (sys.argv[2], k, n[k], tot[k] / n[k]))
PY
    else
        echo "$c: no output ($(tail -1 $OUT/$c.err))"
    fi
done
