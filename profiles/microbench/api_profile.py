"""cProfile of the reference's one-game-at-a-time API on the GPU: GameControl.start_self_play + AlphaZeroPlayer(agent.policy_value_fn)."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rlzero.games.gomoku import GameControl, GomokuEnv
from rlzero.games.gomoku.alphazero_agent import AlphaZeroAgent
from rlzero.mcts.alphazero_mcts import AlphaZeroPlayer
for B, n, sims in ((3, 3, 25), (6, 4, 400)):
    torch.manual_seed(0); np.random.seed(0)
    agent = AlphaZeroAgent(B, device='cuda:0')
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=sims, c_puct=5, is_selfplay=True)
    game = GameControl(GomokuEnv(B, n))
    for _ in range(3): list(game.start_self_play(player, temperature=1.0)[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); plies = 0
    for _ in range(20):
        w, data = game.start_self_play(player, temperature=1.0); plies += len(list(data))
    dt = time.perf_counter() - t0
    print('%dx%d %d sims: %.3f ms per move, %.0f sims/s' % (B, B, sims, 1e3 * dt / plies, plies * sims / dt))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): list(game.start_self_play(player, temperature=1.0)[1])
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
    player.mcts._engine.close()
