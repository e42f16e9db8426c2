#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02d; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
timeout -k 10 300 $B --lanes 1 --games 512 > $OUT/lit_1lane.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 0 > $OUT/lit_2lanes_uncapped.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 0 --heads-algo split64 > $OUT/lit_2lanes_uncapped_h64.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 0 --heads-algo f32 > $OUT/lit_2lanes_uncapped_hf32.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 240 > $OUT/lit_2lanes_240.json 2>/dev/null
timeout -k 10 300 $B --lanes 3 --games 513 --trunk-wgs 0 > $OUT/lit_3lanes_uncapped.json 2>/dev/null
timeout -k 10 300 $B --lanes 4 --games 512 --trunk-wgs 0 > $OUT/lit_4lanes_uncapped.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 1536 --trunk-wgs 0 > $OUT/big_2lanes_1536_uncapped.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 1024 --trunk-wgs 0 > $OUT/big_2lanes_1024_uncapped.json 2>/dev/null
timeout -k 10 300 $B --lanes 3 --games 1536 --trunk-wgs 0 > $OUT/big_3lanes_1536_uncapped.json 2>/dev/null
timeout -k 10 300 $B > $OUT/default_1344.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02d/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}; sk=r.get('small_kernels') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), rf.get('exclusive_launch_ms'), sk)
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
