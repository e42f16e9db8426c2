#!/bin/bash
# round 2, first GPU call: the whole gpu test tier, then the default bench line and a few lane sweeps of the literal config
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
timeout -k 10 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench rc=$?"
for wgs in 112 128 160 192; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --lanes 2 --games 512 --trunk-wgs $wgs > "$OUT/lit_2lanes_wgs$wgs.json" 2>/dev/null
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --lanes 1 --games 512 > "$OUT/lit_1lane.json" 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --lanes 3 --games 512 --trunk-wgs 86 > "$OUT/lit_3lanes_wgs86.json" 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --lanes 4 --games 512 --trunk-wgs 64 > "$OUT/lit_4lanes_wgs64.json" 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r02a/*.json'))):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), r['value'], r['ms_per_step'], (r.get('roofline') or {}).get('frac'))
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
