#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02e; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heads or fused or hip_net or selfplay or lanes" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
timeout -k 10 300 $B --lanes 1 --games 512 > $OUT/lit_1lane_auto.json 2>/dev/null
timeout -k 10 300 $B --lanes 1 --games 512 --heads-algo parts > $OUT/lit_1lane_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 512 --trunk-wgs 0 --heads-algo parts > $OUT/lit_2lanes_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 3 --games 513 --trunk-wgs 0 --heads-algo parts > $OUT/lit_3lanes_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 4 --games 512 --trunk-wgs 0 --heads-algo parts > $OUT/lit_4lanes_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 1024 --trunk-wgs 0 --heads-algo parts > $OUT/big_2lanes_1024_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 2 --games 1536 --trunk-wgs 0 --heads-algo parts > $OUT/big_2lanes_1536_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 3 --games 1536 --trunk-wgs 0 --heads-algo parts > $OUT/big_3lanes_1536_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 3 --games 768 --trunk-wgs 0 --heads-algo parts > $OUT/big_3lanes_768_parts.json 2>/dev/null
timeout -k 10 300 $B --lanes 4 --games 1024 --trunk-wgs 0 --heads-algo parts > $OUT/big_4lanes_1024_parts.json 2>/dev/null
timeout -k 10 300 $B --heads-algo parts > $OUT/default_1344_parts.json 2>/dev/null
timeout -k 10 300 $B > $OUT/default_1344.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02e/*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r.get('roofline') or {}; sk=r.get('small_kernels') or {}
        print(os.path.basename(f), r['value'], r['ms_per_step'], rf.get('frac'), rf.get('avg_launch_ms'), rf.get('exclusive_launch_ms'), sk)
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
