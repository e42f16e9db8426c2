#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r02ah; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o s -- $B --graph 0 --board 9 --playouts 200 --games 64 --lanes 1 --steps 2 --warmup 1 > $OUT/c2.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c1 -o s -- $B --graph 0 --board 3 --playouts 25 --games 1 --lanes 1 --steps 9 --warmup 5 > $OUT/c1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2g -o s -- $B --board 9 --playouts 200 --games 64 --lanes 1 --steps 4 --warmup 2 > $OUT/c2g.json 2>/dev/null
for d in c2 c1 c2g; do echo "== $d"; f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv,sys
for row in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print('%-60s calls %7s avg %9.1f ns  %5s%%' % (row['Name'][:60], row['Calls'], float(row['AverageNs']), row['Percentage']))
PY
done
