#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02b
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_multi_sim.py tests/test_multi_gpu_gloo.py -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
tail -25 "$OUT/pytest_gpu.log"
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
for K in 1 2 4 8 16; do
  timeout -k 10 300 $B --board 9 --playouts 200 --games 64 --lanes 1 --steps 8 --warmup 2 --in-flight $K > "$OUT/c2_1lane_K$K.json" 2>"$OUT/c2_1lane_K$K.err"
done
for K in 4 8; do
  timeout -k 10 300 $B --board 9 --playouts 200 --games 64 --lanes 2 --trunk-wgs 224 --steps 8 --warmup 2 --in-flight $K > "$OUT/c2_2lanes_K$K.json" 2>/dev/null
  timeout -k 10 300 $B --board 9 --playouts 200 --games 64 --lanes 4 --trunk-wgs 224 --steps 8 --warmup 2 --in-flight $K > "$OUT/c2_4lanes_K$K.json" 2>/dev/null
done
for K in 1 5; do
  timeout -k 10 300 $B --board 3 --playouts 25 --games 1 --lanes 1 --steps 9 --warmup 2 --in-flight $K > "$OUT/c1_K$K.json" 2>/dev/null
done
timeout -k 10 300 $B --game muzero --playouts 50 --games 4096 --steps 8 --warmup 2 > "$OUT/c5.json" 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r02b/*.json'))):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}; sk=r.get('small_kernels') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), sk)
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
