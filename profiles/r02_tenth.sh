#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r02j; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --no-configs"
timeout -k 10 300 $B --board 3 --playouts 25 --games 1 --lanes 1 --steps 9 --warmup 2 > $OUT/c1_batched.json 2>/dev/null
python - <<'PY' > $OUT/c1_reference_api.txt 2>&1
import time, numpy as np, torch
from rlzero_amd.games import GameControl, GomokuEnv
from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
from rlzero_amd.mcts import AlphaZeroPlayer
for B, n, sims, K in ((3, 3, 25, 1), (3, 3, 25, 5), (6, 4, 400, 1), (15, 5, 800, 1), (15, 5, 800, 8)):
    torch.manual_seed(0); np.random.seed(0)
    agent = AlphaZeroAgent(B, device='cuda:0')
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=sims, c_puct=5, is_selfplay=True, sims_in_flight=K)
    env = GomokuEnv(B, n)
    GameControl(env).start_self_play(player, temperature=1.0)
    t0 = time.perf_counter(); moves = 0
    for _ in range(3 if B < 15 else 1):
        winner, data = GameControl(env).start_self_play(player, temperature=1.0)
        moves += len(list(data))
    dt = time.perf_counter() - t0
    print('reference API %dx%d %d sims/move, %d in flight: %.1f k sims/s (%d moves, %.1f ms per move)' % (B, B, sims, K, moves * sims / dt / 1e3, moves, 1e3 * dt / moves), flush=True)
PY
cat $OUT/c1_reference_api.txt
