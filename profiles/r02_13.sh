#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02m; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
timeout -k 10 300 $B > $OUT/default.json 2>/dev/null
timeout -k 10 300 $B --lanes 1 --games 512 > $OUT/lit_1lane.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 0 --heads-algo parts > $OUT/lit_2lanes_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 1536 --trunk-wgs 0 --heads-algo parts > $OUT/big_2lanes_1536_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 1 --games 512 --noise 0 > $OUT/lit_1lane_nonoise.json 2>/dev/null
timeout -k 10 300 $B --game connect4 --playouts 400 --games 512 --lanes 1 --steps 6 --warmup 2 > $OUT/c3.json 2>/dev/null
timeout -k 10 300 $B --board 9 --playouts 200 --games 64 --lanes 1 --steps 8 --warmup 2 --in-flight 16 > $OUT/c2_K16.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02m/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}; sk=r.get('small_kernels') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), sk, r.get('warmup_moves_run'))
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
