#!/bin/bash
# The deferred policy GEMM (k_heads_split over the store) after pairing its workgroups per XCD: duration and HBM reads per launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/rz_gemm; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0 --graph 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B --steps 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_heads_split' in r['Name'] or 'k_deferred_priors' in r['Name']: print(r['Name'][:60], 'calls', r['Calls'], 'avg us', float(r['AverageNs']) / 1e3)
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob('$OUT/pmc_$c/**/*counter_collection.csv', recursive=True)[0]
tot = {}; n = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'k_heads_split' in k:
        tot[k[:40]] = tot.get(k[:40], 0.0) + float(r['Counter_Value']); n[k[:40]] = n.get(k[:40], 0) + 1
for k in tot: print('$c', k, 'per launch (KB as counted)', tot[k] / n[k], 'launches', n[k])
PY
done
