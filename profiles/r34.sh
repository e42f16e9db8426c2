#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02af; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_muzero.py -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
python profiles/tmp_prof/run.py 2>&1 | grep -v amdgpu.ids
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --game muzero"
for G in 4096 8192; do
  timeout -k 10 300 $B --games $G --steps 128 --warmup 16 > $OUT/mz_${G}.json 2>$OUT/mz_${G}.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02af/mz_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(r['value']/1e6,2), r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline'].get('moves_per_launch'))
    except Exception as e: print(os.path.basename(f),'ERR',e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
