#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02n; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --steps 6 --warmup 2"
for rep in 1 2 3; do
  for noise in 1 0; do
    timeout -k 10 300 $B --lanes 1 --games 512 --noise $noise > $OUT/lit_1lane_noise${noise}_$rep.json 2>/dev/null
  done
done
for rep in 1 2; do
  timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 0 --heads-algo parts > $OUT/lit_2lanes_parts_$rep.json 2>/dev/null
  timeout -k 10 300 $B > $OUT/default_$rep.json 2>/dev/null
  timeout -k 10 300 $B --lanes 2 --games 1536 --trunk-wgs 0 --heads-algo parts > $OUT/big_1536_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02n/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), rf.get('exclusive_launch_ms'))
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
