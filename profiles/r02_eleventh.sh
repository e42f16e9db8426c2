#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02k; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_muzero.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
timeout -k 10 300 $B --game muzero --playouts 50 --games 4096 --steps 16 --warmup 4 > $OUT/c5_fused.json 2>$OUT/c5_fused.err
timeout -k 10 300 $B --game muzero --playouts 50 --games 4096 --steps 16 --warmup 4 --mz-fused 0 > $OUT/c5_graph.json 2>/dev/null
timeout -k 10 300 $B --game muzero --playouts 50 --games 16384 --steps 16 --warmup 4 > $OUT/c5_fused_16k.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02k/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), rf.get('achieved'), rf.get('unit'))
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
tail -3 $OUT/c5_fused.err
