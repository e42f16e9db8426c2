#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02l; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_multi_sim.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout -k 10 300 python __graft_entry__.py --smoke 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
for K in 4 8 16; do
  timeout -k 10 300 $B --board 9 --playouts 200 --games 64 --lanes 1 --steps 8 --warmup 2 --in-flight $K > $OUT/c2_K$K.json 2>/dev/null
done
timeout -k 10 300 $B --board 15 --playouts 800 --games 64 --lanes 1 --steps 4 --warmup 1 --in-flight 8 > $OUT/c4_64games_K8.json 2>/dev/null
timeout -k 10 300 $B --board 15 --playouts 800 --games 64 --lanes 1 --steps 4 --warmup 1 --in-flight 1 > $OUT/c4_64games_K1.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02l/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}; sk=r.get('small_kernels') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), sk)
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
