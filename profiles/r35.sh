#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02ag; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs --game muzero"
for rep in 1 2 3 4; do
  timeout -k 10 300 $B --games 8192 --steps 256 --warmup 48 > $OUT/mz_8192_$rep.json 2>$OUT/mz_8192_$rep.err
done
timeout -k 10 300 $B --games 16384 --steps 256 --warmup 48 > $OUT/mz_16384_1.json 2>$OUT/mz_16384_1.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02ag/mz_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(r['value']/1e6,2), r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline'].get('moves_per_launch'))
    except Exception as e: print(os.path.basename(f),'ERR',e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
python - <<'PY'
# where the host's time per launch goes
import time, torch, numpy as np
from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
torch.manual_seed(0)
net = MuZeroNet().to('cuda').eval()
sp = MuZeroSelfPlay(net, CartPoleBatch(8192, torch.device('cuda'), seed=0), n_sims=50, seed=0)
sp.collect(64)
bufs = sp._fused_state()
for i in range(4):
    t0 = time.perf_counter(); sp._launch_moves(16, bufs[0]); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eps = sp._episodes_of_launch(bufs[0]); t3 = time.perf_counter()
    print('launch call %.2f ms, gpu+copy %.2f ms, host episodes %.2f ms (%d episodes, %d rows)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(eps), int(eps.lengths().sum())))
PY
