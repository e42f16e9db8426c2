"""The PRODUCTION tree kernels against the oracle, directly, at the BASELINE sizes.

The synthetic-evaluator goldens (G2 / G3) run through k_select + k_expand_backup; every real-net run goes through other kernels:
the resident search (rz_net_search_resident: k_trunk_rows / k_trunk_split <RES>, one launch per search) and the two-launch step
(k_trunk_* + k_tree_step_def, replayed from hipGraphs of 16 steps).  Here those routes meet the oracle's sequential search
(rlzero/mcts/node.py:32-88,119-144, rlzero/mcts/alphazero_mcts.py:42-103) with no chain of route-equals-route arguments in
between: C1 (3 x 3, 25 simulations), C2 (9 x 9, 200), C4 (15 x 15, 800; both routes), C3 (Connect4, 400), several games, four
moves with tree reuse (moves drawn from injected uniforms the way numpy.random.choice draws them), Dirichlet noise on.

The oracle is fed, leaf by leaf, the value the device gives that position (a one-game engine expanding the position as its
root: the value of a board does not depend on its batch, route or slot -- which is part of what is asserted): after every move
every visited node's N and W (fp64 bits) and pi (1e-12) are the oracle's.  The second test plays a whole 6 x 6 game through the
reference's own entry points -- GameControl.start_self_play + AlphaZeroPlayer(agent.policy_value_fn) with the agent on the GPU
(alphazero_mcts.py:136-165, game.py:96-134) -- against the oracle's player on the same values and uniforms."""
import numpy as np
import pytest

from oracle.connect4_ref import RefConnect4
from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import RefPlayer, RefSearch, inverse_cdf_choice, self_play_game, tree_dump

pytestmark = pytest.mark.gpu


def _hex_tree(d):
    return {k: (n, float(w).hex()) for k, (n, w) in d.items()}


def _last(env):
    return getattr(env, 'last_cell', env.last_move)


def _set_roots(eng, envs):
    from rlzero_amd.engine import int_to_bits
    stones = np.array([[int_to_bits(e.bitboards()[0]), int_to_bits(e.bitboards()[1])] for e in envs], dtype=np.uint64)
    eng.set_roots(stones, [e.current_player() for e in envs], [_last(e) for e in envs], reset_trees=True)


class _Probe(object):
    """policy_value_fn for the oracle: the leaf value the device's evaluator gives a position, read from a one-game engine that
    expands the position as its root (W(root) = -value after one simulation); priors uniform (the reference's rule never reads
    them).  Terminal positions are not evaluated: the reference discards the value there (alphazero_mcts.py:59-68)."""

    def __init__(self, net, game, shape, n_row):
        from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
        net_shape = (shape[0], shape[1], shape[1]) if game == 'connect4' else shape
        self.evaluator = HipNetEvaluator(net, net_shape, 'cuda:0', max_boards=1)
        self.eng = MCTSEngine(shape, n_row, n_games=1, n_playout=4, device='cuda:0', game=game)
        self.cache, self.calls = {}, 0

    def __call__(self, env):
        legal = env.leagel_actions()
        priors = [(a, 1.0 / len(legal)) for a in legal]
        if env.game_end_winner()[0]:
            return priors, 0.0
        key = (env.bitboards(), env.current_player(), _last(env))
        if key not in self.cache:
            _set_roots(self.eng, [env])
            self.eng.sim_chunk(self.evaluator, 1)
            self.cache[key] = -float(self.eng.root_stats()[1][0])
            self.calls += 1
        return priors, self.cache[key]

    def close(self):
        self.eng.close()
        self.evaluator.hip.close()


def _start_positions(game, shape, n_row, count, seed):
    rs = np.random.RandomState(seed)
    if game == 'connect4':
        envs = [RefConnect4(shape[0], shape[1], n_row), RefConnect4.from_moves([3, 3], shape[0], shape[1], n_row),
                RefConnect4.from_moves([3, 3, 2, 4, 2, 4], shape[0], shape[1], n_row)]   # three in a column soon: terminal leaves
        make = lambda: RefConnect4(shape[0], shape[1], n_row)   # noqa: E731
        cells = shape[0] * shape[1]
    else:
        B = shape
        envs = [RefGomoku(B, n_row), RefGomoku.from_moves(B, n_row, [B * B // 2])]
        if n_row >= 7:   # six stones of player 0 on the main diagonal, to move: a seventh makes NO line of eight (a window of n cells that
            # spans more than 64 cell numbers is tested cell by cell, not by the 64-bit mask of shorter windows: rz_tree.h, line_through)
            envs.append(RefGomoku.from_moves(B, n_row, [0, 1, B + 1, 3, 2 * (B + 1), 5, 3 * (B + 1), 7, 4 * (B + 1), 2 * B - 1, 5 * (B + 1), 3 * B - 1]))
        if B >= 9:   # an open three in a row of n - 1 next to the centre: wins, losses and terminal leaves inside the search
            c = B * (B // 2) + B // 2
            envs.append(RefGomoku.from_moves(B, n_row, [c, c + B, c + 1, c + B + 1, c + 2, c + B + 2][:2 * (n_row - 2)]))
        make = lambda: RefGomoku(B, n_row)   # noqa: E731
        cells = B * B
    while len(envs) < count:
        e = make()
        for _ in range(rs.randint(1, max(2, cells // 2))):
            e.step(int(rs.choice(e.leagel_actions())))
            if e.game_end_winner()[0]:
                break
        if not e.game_end_winner()[0] and len(e.leagel_actions()) >= 4:
            envs.append(e)
    return envs[:count]


def _near_wins(B, n_row, short=1):
    """Four roots, one per direction of gomoku_env.py:136-168 (right, down, down-right, down-left): the player to move has n - 1 stones
    in that direction from a corner of the board and the completing cell is free -- every search from them meets a terminal leaf at
    depth one (and, among its neighbours, windows of n - 1 stones that are NOT wins).  short = 2: n - 2 stones -- the search completes
    windows of n - 1, which must NOT end the game (a window test that loses its last cell would say they do)."""
    out = []
    for dy, dx, y0, x0 in ((0, 1, 0, 0), (1, 0, 0, B - 1), (1, 1, B - n_row, B - n_row), (1, -1, 0, n_row - 1)):
        mine = [(y0 + j * dy) * B + (x0 + j * dx) for j in range(n_row - short)]
        last = (y0 + (n_row - short) * dy) * B + (x0 + (n_row - short) * dx)
        taken, theirs = set((y0 + j * dy) * B + (x0 + j * dx) for j in range(n_row)), []
        for c in range(B * B - 1, -1, -1):   # the opponent's stones: from the far end, off the line, no two side by side
            if len(theirs) == n_row - short:
                break
            if c not in taken and (c + 1) not in theirs and (c % 3 == 2 or B < 5):
                theirs.append(c)
        assert len(theirs) == n_row - short
        moves = [m for pair in zip(mine, theirs) for m in pair]
        env = RefGomoku.from_moves(B, n_row, moves)
        assert not env.game_end_winner()[0] and last in env.leagel_actions()
        out.append(env)
    return out


def _draw(acts, probs, u):
    cdf = np.cumsum(np.asarray(probs, dtype=np.float64))
    cdf /= cdf[-1]
    i = int(cdf.searchsorted(u, side='right'))
    edges = np.concatenate([[0.0], cdf])
    assert min(abs(u - edges[i]), abs(edges[i + 1] - u)) > 1e-9   # (no uniform on a cdf edge)
    return int(acts[i])


CASES = [   # (id, game, shape, n_in_row, simulations per move, routes, games)
    ('C1_3x3_25', 'gomoku', 3, 3, 25, ('resident', 'two_launch_graph'), 8),
    ('C2_9x9_200', 'gomoku', 9, 5, 200, ('resident', 'two_launch_graph'), 6),
    ('C4_15x15_800', 'gomoku', 15, 5, 800, ('resident', 'two_launch_graph'), 6),
    ('C3_connect4_400', 'connect4', (6, 7), 4, 400, ('two_launch_graph', 'resident'), 5),
    # the tree code is compiled per number of 64-bit words a board's bitboards use (rz_tree.h: W) -- the boundaries: 8 x 8 = 64 cells
    # is the last one-word board (cell 63 = bit 63), 10 x 10 the largest two-word board of k_trunk_split, 11 x 11 = 121 cells runs the
    # two-word tree step beside the row kernel's trunk (and the four-word code inside the resident row kernel)
    ('W1_8x8_120', 'gomoku', 8, 5, 120, ('resident', 'two_launch_graph'), 4),
    ('W2_10x10_120', 'gomoku', 10, 5, 120, ('resident', 'two_launch_graph'), 4),
    ('W2_11x11_120', 'gomoku', 11, 5, 120, ('resident', 'two_launch_graph'), 4),
    # n = 8 on 9 x 9: a diagonal window spans 7 x 10 = 70 cell numbers
    ('N8_9x9_150', 'gomoku', 9, 8, 150, ('resident', 'two_launch_graph'), 4),
    # the other shapes of the window test: 10 x 10 with n = 7 (a high mask word), 8 x 8 with n = 8 (one word, the longest window),
    # 16 x 16 (the widest board whose first n - 1 cells still fit the mask), 15 x 15 with n = 8 (they do not: cell by cell)
    ('N7_10x10_80', 'gomoku', 10, 7, 80, ('resident', 'two_launch_graph'), 4),
    ('N8_8x8_80', 'gomoku', 8, 8, 80, ('resident', 'two_launch_graph'), 4),
    ('N5_16x16_80', 'gomoku', 16, 5, 80, ('resident', 'two_launch_graph'), 4),
    ('N8_15x15_80', 'gomoku', 15, 8, 80, ('resident', 'two_launch_graph'), 4),
    # the same shapes from roots one move short of a line in each of the four directions (`near`): terminal leaves in every search
    ('near_3x3_n3', 'gomoku', 3, 3, 25, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_6x6_n4', 'gomoku', 6, 4, 60, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_8x8_n8', 'gomoku', 8, 8, 60, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_9x9_n5', 'gomoku', 9, 5, 100, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_9x9_n8', 'gomoku', 9, 8, 100, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_10x10_n7', 'gomoku', 10, 7, 100, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_11x11_n5', 'gomoku', 11, 5, 100, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_15x15_n5', 'gomoku', 15, 5, 250, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_15x15_n8', 'gomoku', 15, 8, 250, ('resident', 'two_launch_graph'), 4, 'near'),
    ('near_16x16_n5', 'gomoku', 16, 5, 270, ('resident', 'two_launch_graph'), 4, 'near'),
    ('short_9x9_n8', 'gomoku', 9, 8, 100, ('resident', 'two_launch_graph'), 4, 'short'),
    ('short_10x10_n7', 'gomoku', 10, 7, 100, ('resident', 'two_launch_graph'), 4, 'short'),
    ('short_11x11_n7', 'gomoku', 11, 7, 130, ('resident', 'two_launch_graph'), 4, 'short'),
    ('short_15x15_n8', 'gomoku', 15, 8, 250, ('resident', 'two_launch_graph'), 4, 'short'),
    ('short_16x16_n5', 'gomoku', 16, 5, 270, ('resident', 'two_launch_graph'), 4, 'short'),
]
N_MOVES = 4


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_production_routes_are_the_oracle_s_search(case):
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import visits_to_pi
    _, game, shape, n_row, sims, routes, n_games = case[:7]
    torch.manual_seed(7)
    net = PolicyValueNet(6, 7, 7) if game == 'connect4' else PolicyValueNet(shape)
    net_shape = (6, 7, 7) if game == 'connect4' else shape
    probe = _Probe(net, game, shape, n_row)
    starts = (_near_wins(shape, n_row, 2 if case[7] == 'short' else 1) if len(case) > 7
              else _start_positions(game, shape, n_row, n_games, seed=sims))
    uniforms = np.random.RandomState(sims + 1).random_sample((n_games, N_MOVES))

    # the oracle: per game N_MOVES searches with tree reuse; per move the whole tree, pi and the move drawn from it
    want = []
    for g, start in enumerate(starts):
        env, ref, per_move = start.clone(), RefSearch(probe, sims, 5), []
        for m in range(N_MOVES):
            if env.game_end_winner()[0]:
                break
            acts, probs = ref.simulate(env, 1.0)
            move = _draw(acts, probs, uniforms[g, m])
            per_move.append((_hex_tree(tree_dump(ref.root)), tuple(acts), probs, move))
            ref.update_with_move(move)   # alphazero_mcts.py:96-103
            env.step(move)
        want.append(per_move)
    assert probe.calls > sims   # (the values really came from the device)

    for route in routes:
        evaluator = HipNetEvaluator(net, net_shape, 'cuda:0', max_boards=n_games)
        evaluator.resident_search = route == 'resident'
        eng = MCTSEngine(shape, n_row, n_games=n_games, n_playout=sims, device='cuda:0', game=game, add_noise=True, noise_seed=11)
        assert evaluator.deferred_ok(eng) and evaluator.resident_ok(eng) == (route == 'resident')
        graph = route == 'two_launch_graph'
        if graph:
            eng.reset_games()
            eng.warm_graph(evaluator, 16)
        _set_roots(eng, starts)
        eng.set_noise_keys()
        active = np.ones(n_games, dtype=np.uint8)
        for m in range(N_MOVES):
            for g in range(n_games):
                if len(want[g]) <= m:
                    active[g] = 0   # the game ended with the last move
            if not active.any():
                break
            eng.set_active(active)
            eng.simulate(evaluator, sims, use_graph=graph, sims_per_graph=16)
            visits = eng.root_visits()
            moves = np.full(n_games, -2, dtype=np.int32)
            for g in range(n_games):
                if not active[g]:
                    continue
                tree, acts, probs, move = want[g][m]
                assert _hex_tree(eng.tree_dump(g)) == tree, (route, 'game %d move %d' % (g, m))
                pi = visits_to_pi(visits[g][list(acts)], 1.0)
                assert np.max(np.abs(pi - probs)) <= 1e-12
                assert _draw(acts, pi, uniforms[g, m]) == move
                moves[g] = move
            eng.advance(moves)   # tree reuse before the boards change
            eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
        st = eng.check()
        evaluator.hip.check_flags()
        assert st.reuse_dropped == 0
        eng.close()
        evaluator.hip.close()
    probe.close()


def test_whole_game_through_the_reference_api_with_a_gpu_agent():
    """GameControl.start_self_play(AlphaZeroPlayer(agent.policy_value_fn, is_selfplay=True)) with the agent on the GPU -- the
    fast path: the player's one-game engine searched by the resident kernel -- for a whole 6 x 6 game at the script's defaults
    (four in a row, 400 playouts, train_alphazero.py:21-31) == the oracle's player and game loop fed the device's values and the
    same uniforms: moves, pi (1e-12), z, winner and the observation planes."""
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games.gomoku import GameControl, GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts.alphazero_mcts import AlphaZeroPlayer
    import torch
    B, n_row, sims = 6, 4, 400
    torch.manual_seed(5)
    agent = AlphaZeroAgent(B, device='cuda:0')
    us = np.random.RandomState(17).random_sample(B * B).tolist()
    queue = list(us)

    def injected(acts, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return np.asarray(acts)[cdf.searchsorted(queue.pop(0), side='right')]

    real = np.random.choice
    np.random.choice = injected
    try:
        env = GomokuEnv(B, n_row)
        player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=sims, c_puct=5, is_selfplay=True)
        winner, data = GameControl(env).start_self_play(player, temperature=1.0)
        data = list(data)
    finally:
        np.random.choice = real
    evaluator = player.mcts._evaluator
    assert isinstance(evaluator, HipNetEvaluator) and evaluator.resident_ok(player.mcts._engine)
    moves = list(env.states.keys())

    probe = _Probe(agent.policy_value_net, 'gomoku', B, n_row)
    ref_player = RefPlayer(probe, sims, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
    w2, data2, moves2 = self_play_game(RefGomoku(B, n_row), ref_player, temperature=1.0)
    assert winner == w2 and moves == moves2 and len(data) == len(data2) >= 2 * n_row - 1
    for (s1, p1, z1), (s2, p2, z2) in zip(data, data2):
        assert (s1 == s2).all() and np.max(np.abs(p1 - p2)) <= 1e-12 and z1 == z2
    assert player.mcts._engine.check().reuse_dropped == 0
    player.mcts._engine.close()
    probe.close()
