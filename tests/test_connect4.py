"""Connect4 (BASELINE.json config 3).  The reference has no Connect4, so parity is pinned against
this repository's naive twin (oracle/connect4_ref.py) and by rule properties."""
import numpy as np
import pytest

from oracle import evaluators as ev
from oracle.connect4_ref import RefConnect4
from oracle.mcts_ref import RefPlayer, RefSearch, inverse_cdf_choice, self_play_game, tree_dump


def _random_game(rs, rows=6, cols=7, n=4):
    env = RefConnect4(rows, cols, n)
    moves = []
    while not env.game_end_winner()[0]:
        a = int(rs.choice(env.leagel_actions()))
        env.step(a)
        moves.append(a)
    return moves, env.game_end_winner()[1]


def test_product_env_equals_naive_twin():
    from rlzero_amd.games import Connect4Env
    rs = np.random.RandomState(4)
    outcomes = set()
    for shape in ((6, 7, 4), (6, 7, 4), (5, 5, 3), (4, 9, 4), (6, 7, 4)):
        for _ in range(12):
            ref, env = RefConnect4(*shape), Connect4Env(*shape)
            while True:
                assert env.leagel_actions() == ref.leagel_actions()
                assert (env.current_state() == ref.current_state()).all()
                assert env.game_end_winner() == ref.game_end_winner()
                assert env.bitboards() == ref.bitboards() and env.states == ref.states
                if ref.game_end_winner()[0]:
                    outcomes.add(ref.game_end_winner()[1])
                    break
                a = int(rs.choice(ref.leagel_actions()))
                obs, reward, win, _ = env.step(a)
                ref.step(a)
                assert (env.last_move, env.last_cell) == (ref.last_move, ref.last_cell)
            twin = Connect4Env.from_bitboards(shape[0], shape[1], shape[2], *env.bitboards(),
                                              env.current_player(), env.last_cell)
            assert (twin.current_state() == env.current_state()).all()
            assert twin.leagel_actions() == env.leagel_actions() and twin.game_end_winner() == env.game_end_winner()
    assert outcomes >= {0, 1}


def test_rule_properties():
    env = RefConnect4()
    for a in (3, 3, 4, 4, 5, 5):
        env.step(a)
    assert env.game_end_winner() == (False, -1)
    env.step(6)  # player 0 completes 3,4,5,6 on the bottom row
    assert env.game_end_winner() == (True, 0)
    env = RefConnect4.from_moves([0, 1, 0, 1, 0, 1, 0])  # vertical four in column 0
    assert env.game_end_winner() == (True, 0) and env.cells[21] == 0
    env = RefConnect4.from_moves([0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3])  # rising diagonal (0,0)..(3,3)
    assert env.game_end_winner() == (True, 0)
    env = RefConnect4.from_moves([0] * 6)
    assert env.leagel_actions() == [1, 2, 3, 4, 5, 6]
    with pytest.raises(AssertionError):
        env.step(0)
    # gravity: stones stack, cell = row*7 + col
    env = RefConnect4.from_moves([2, 2, 2])
    assert sorted(env.states) == [2, 9, 16] and env.last_cell == 16


pytestmark_gpu = pytest.mark.gpu


def _engine(**kw):
    from rlzero_amd.engine import MCTSEngine
    return MCTSEngine((6, 7), 4, game='connect4', **kw)


def _hex_tree(dump):
    return {p: (n, float(w).hex()) for p, (n, w) in dump.items()}


@pytest.mark.gpu
def test_connect4_rules_on_device():
    """rz_step_games with column actions + gravity vs the naive twin, 64 random games in lock-step."""
    rs = np.random.RandomState(8)
    games = [_random_game(rs) for _ in range(64)]
    eng = _engine(n_games=64, n_playout=4)
    eng.reset_games()
    envs = [RefConnect4() for _ in games]
    for ply in range(max(len(m) for m, _ in games)):
        moves = np.array([m[ply] if ply < len(m) else -1 for m, _ in games], dtype=np.int32)
        winner, ended = eng.step(moves)
        obs = eng.root_obs().cpu().numpy()
        for g, (m, w) in enumerate(games):
            if ply < len(m):
                envs[g].step(m[ply])
                assert (bool(ended[g]), int(winner[g])) == envs[g].game_end_winner()
                assert (obs[g] == envs[g].current_state()).all()
    eng.check()
    eng.step(np.array([0] * 64, dtype=np.int32))  # at least one finished game has column 0 full or not: flags only if full
    eng.close()


@pytest.mark.gpu
def test_connect4_search_vs_oracle():
    """UCT_REF search with column actions on 32 random positions + PUCT on a few, device vs twin."""
    from rlzero_amd.engine import HostEvaluator, SyntheticEvaluator, int_to_bits
    from rlzero_amd.games import Connect4Env
    rs = np.random.RandomState(12)
    envs = []
    while len(envs) < 32:
        env = RefConnect4()
        for _ in range(rs.randint(0, 30)):
            if env.game_end_winner()[0]:
                break
            env.step(int(rs.choice(env.leagel_actions())))
        if not env.game_end_winner()[0]:
            envs.append(env)
    eng = _engine(n_games=32, n_playout=300)
    stones = np.array([[int_to_bits(e.bitboards()[0]), int_to_bits(e.bitboards()[1])] for e in envs], dtype=np.uint64)
    eng.set_roots(stones, [e.current_player() for e in envs], [e.last_cell for e in envs], reset_trees=True)
    eng.simulate(SyntheticEvaluator('vlin'), 300)
    visits, wsum = eng.root_visits(), eng.root_wsum()
    eng.check()
    assert visits.shape == (32, 7)
    for g, env in enumerate(envs):
        s = RefSearch(ev.vlin, 300, 5)
        acts, _ = s.simulate(env, 1.0)
        assert list(acts) == env.leagel_actions()
        assert [int(visits[g, a]) for a in acts] == [k.n for k in s.root.kids]
        assert [float(wsum[g, a]).hex() for a in acts] == [float(k.w).hex() for k in s.root.kids]
        assert _hex_tree(eng.tree_dump(g)) == _hex_tree(tree_dump(s.root))
    eng.close()

    def skewed(env):
        legal = env.leagel_actions()
        raw = np.array([1 + (3 * a + 1) % 4 for a in legal], dtype=np.float32)
        return list(zip(legal, raw / np.float32(raw.sum()))), ev.vlin_value(env.states, env.current_player())

    for env in envs[:3]:
        eng = _engine(n_games=1, n_playout=200, score_mode='puct')
        eng.set_roots(np.array([[int_to_bits(env.bitboards()[0]), int_to_bits(env.bitboards()[1])]], dtype=np.uint64),
                      [env.current_player()], [env.last_cell], reset_trees=True)
        host = HostEvaluator(skewed, lambda s0, s1, tm, last: Connect4Env.from_bitboards(6, 7, 4, s0, s1, tm, last))
        eng.simulate(host, 200)
        eng.check()
        s = RefSearch(skewed, 200, 5, score_mode='puct')
        s.simulate(env, 1.0)
        assert _hex_tree(eng.tree_dump(0)) == _hex_tree(tree_dump(s.root))
        eng.close()


@pytest.mark.gpu
def test_connect4_selfplay_through_reference_api_vs_oracle():
    """GameControl.start_self_play + AlphaZeroPlayer on a Connect4Env == the oracle's self-play."""
    from conftest import unhex  # noqa: F401
    from rlzero_amd.games import Connect4Env, GameControl
    from rlzero_amd.mcts import AlphaZeroPlayer
    us = np.random.RandomState(3).random_sample(64).tolist()
    real = np.random.choice
    queue = list(us)

    def injected(acts, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return np.asarray(acts)[cdf.searchsorted(queue.pop(0), side='right')]

    np.random.choice = injected
    try:
        env = Connect4Env()
        player = AlphaZeroPlayer(ev.vlin, n_playout=120, c_puct=5, is_selfplay=True)
        winner, data = GameControl(env).start_self_play(player, temperature=1.0)
        data = list(data)
    finally:
        np.random.choice = real
    ref_player = RefPlayer(ev.vlin, 120, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
    w2, data2, moves2 = self_play_game(RefConnect4(), ref_player, temperature=1.0)
    assert winner == w2 and [c % 7 for c in env.states.keys()] == moves2 and len(data) == len(data2)
    for (s1, p1, z1), (s2, p2, z2) in zip(data, data2):
        assert (s1 == s2).all() and p1.shape == (7, ) and np.max(np.abs(p1 - p2)) <= 1e-12 and z1 == z2
    player.mcts._engine.close()


@pytest.mark.gpu
def test_connect4_net_and_fast_path():
    """HipNet on the 6x7 board with 7 policy outputs vs torch fp64; the GPU agent's fast path."""
    import torch
    from rlzero_amd.engine import HipNet, HipNetEvaluator
    from rlzero_amd.games import Connect4Env, GameControl
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.mcts import AlphaZeroPlayer
    torch.manual_seed(1)
    net = PolicyValueNet(6, 7, 7)
    hip = HipNet((6, 7, 7), 'cuda:0', max_boards=64).load_state_dict(net.state_dict())
    x = torch.randn(33, 4, 6, 7)
    with torch.no_grad():
        lp64, v64 = net.double()(x.double())
    lp, v = hip.forward(x.to('cuda:0'))
    assert lp.shape == (33, 7)
    assert np.max(np.abs(lp.cpu().numpy() - lp64.numpy())) <= 1e-4
    assert np.max(np.abs(v.cpu().numpy() - v64.numpy()[:, 0])) <= 1e-4
    hip.close()
    np.random.seed(2)
    agent = AlphaZeroAgent(6, device='cuda:0', board_width=7, n_actions=7)
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=60, c_puct=5, is_selfplay=True)
    env = Connect4Env()
    winner, data = GameControl(env).start_self_play(player, temperature=1.0)
    data = list(data)
    assert isinstance(player.mcts._evaluator, HipNetEvaluator)
    assert winner in (-1, 0, 1) and 7 <= len(data) <= 42 and data[0][1].shape == (7, )
    player.mcts._engine.close()


@pytest.mark.gpu
def test_connect4_rollout_vs_oracle():
    from oracle.rollout_ref import RefRolloutSearch
    from rlzero_amd.games import Connect4Env
    from rlzero_amd.mcts.rollout_mcts import RolloutMCTS, rollout_pick
    env = Connect4Env()
    for a in (3, 3, 2, 4):
        env.step(a)
    mcts = RolloutMCTS(n_playout=150, c_puct=5)
    mcts.seed = 21
    move = mcts.simulate(env)
    ref = RefRolloutSearch(150, 5)
    state = {'sim': -1, 'ply': 0}

    def rand(k):
        if ref.sim_index != state['sim']:
            state['sim'], state['ply'] = ref.sim_index, 0
        out = np.zeros(k)
        out[rollout_pick(21, 0, state['sim'], state['ply'], k)] = 1.0
        state['ply'] += 1
        return out

    ref.rand = rand
    assert move == ref.simulate(RefConnect4.from_moves([3, 3, 2, 4]))
    assert _hex_tree(mcts._engine.tree_dump(0)) == _hex_tree(tree_dump(ref.root))
    mcts._engine.close()


@pytest.mark.gpu
def test_connect4_search_routes_agree():
    """Connect4 (6 x 7 cells, 7 column actions: policy outputs != cells) on the three routes of the production evaluator -- the
    resident search (one launch per search), the two-launch step (deferred priors) and the three-launch step: after three moves with
    tree reuse the visit counts, every node's N / W and every prior are the same between the first two, and the third stores the
    same priors for the same first expansion with values equal to f32 rounding."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(3)
    net = PolicyValueNet(6, 7, 7)
    out = {}
    for route in ('resident', 'two_launch', 'three_launch'):
        evaluator = HipNetEvaluator(net, (6, 7, 7), 'cuda:0', max_boards=6)
        evaluator.resident_search = route == 'resident'
        evaluator.deferred_priors = route != 'three_launch'
        eng = _engine(n_games=6, n_playout=70, add_noise=True, noise_seed=4)
        assert evaluator.resident_ok(eng) == (route == 'resident') and evaluator.deferred_ok(eng) == (route != 'three_launch')
        eng.reset_games()
        eng.set_noise_keys()
        eng.sim_chunk(evaluator, 1)
        first = (eng.root_priors().copy(), eng.root_stats()[1].copy())
        eng.sim_chunk(evaluator, 69)
        record = [first]
        for move in range(3):
            if move:
                eng.simulate(evaluator, 70)
            visits = eng.root_visits()
            ar = [eng.arena(g) for g in range(6)]
            record.append((visits.copy(), [(a['N'][0], float(a['W'][0]).hex(), a['PRI'][:7].view(np.uint32).tolist()) for a in ar],
                           [eng.tree_dump(g) for g in range(6)]))
            moves = visits.argmax(axis=1).astype(np.int32)
            eng.advance(moves)
            eng.step(moves)
        eng.check()
        out[route] = record
        eng.close()
        evaluator.hip.close()
    a, b, c = out['resident'], out['two_launch'], out['three_launch']
    assert np.array_equal(a[0][0].view(np.uint32), b[0][0].view(np.uint32)) and np.array_equal(a[0][1], b[0][1])
    assert np.array_equal(a[0][0].view(np.uint32), c[0][0].view(np.uint32)) and np.max(np.abs(a[0][1] - c[0][1])) <= 2e-6
    assert (a[0][0] > 0).sum(axis=1).tolist() == [7] * 6   # seven legal columns on the empty board, a prior each
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x[0], y[0]) and x[1] == y[1]
        assert [{k: (n, float(w).hex()) for k, (n, w) in t.items()} for t in x[2]] == \
            [{k: (n, float(w).hex()) for k, (n, w) in t.items()} for t in y[2]]


@pytest.mark.gpu
def test_connect4_resident_search_on_the_compact_grid_in_rounds():
    """BASELINE configs[2]'s batch and beyond on ONE lane: 600 Connect4 games are a launch of 600 workgroups of the resident search on the
    compact LDS grid, two per CU, the last 88 in a second round.  Visit counts of every game and the trees of a sample equal the two-launch
    step's over two moves with tree reuse, bit for bit; the games start from different openings."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(5)
    net = PolicyValueNet(6, 7, 7)
    G, sims = 600, 40
    rs = np.random.RandomState(2)
    openings = rs.randint(0, 7, size=(2, G)).astype(np.int32)
    out = {}
    for resident in (True, False):
        evaluator = HipNetEvaluator(net, (6, 7, 7), 'cuda:0', max_boards=G)
        evaluator.resident_search = resident
        eng = _engine(n_games=G, n_playout=sims, add_noise=True, noise_seed=8)
        assert evaluator.resident_ok(eng) == resident and evaluator.deferred_ok(eng) and evaluator.resident_per_cu(eng) == 2
        eng.reset_games()
        for ply in range(2):
            eng.advance(openings[ply])   # (nothing searched yet: a fresh root each time)
            eng.step(openings[ply])
        eng.set_noise_keys()
        record = []
        for move in range(2):
            eng.simulate(evaluator, sims, use_graph=False)
            visits = eng.root_visits()
            record.append((visits.copy(), [eng.tree_dump(g) for g in range(0, G, 41)]))
            moves = visits.argmax(axis=1).astype(np.int32)
            eng.advance(moves)
            eng.step(moves)
        eng.check()
        out[resident] = record
        eng.close()
        evaluator.hip.close()
    for x, y in zip(out[True], out[False]):
        assert np.array_equal(x[0], y[0])
        assert [{k: (n, float(w).hex()) for k, (n, w) in t.items()} for t in x[1]] == \
            [{k: (n, float(w).hex()) for k, (n, w) in t.items()} for t in y[1]]
