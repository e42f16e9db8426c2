#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02o; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r02o/bench_default.json').read().strip().splitlines()[-1])
print(r['config']['workload'], r['value'], r['ms_per_step'], r['selfplay_games_per_sec'], r['roofline']['frac'], r['roofline'].get('exclusive_frac'))
print(r['literal_config'])
for k,v in r['configs'].items(): print(k, v.get('value'), (v.get('roofline') or {}).get('frac'), (v.get('cpu_baseline') or {}).get('value'))
print(r['cpu_baseline']['value'])
PY
