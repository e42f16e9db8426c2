"""The opt-in mode with K simulations in flight per tree (virtual loss): NOT the reference's algorithm (its
simulations are strictly sequential, rlzero/mcts/alphazero_mcts.py:82-85) and therefore excluded from every parity
test; what is tested are the properties any correct search keeps: N(root) grows by exactly n_playout, visits are
conserved (N(node) = ended-here + sum N(children)), every virtual loss is gone after the last backup (W(node) +
sum W(children) = -(visits that ended at the node) x value of the node's position, exactly, with the exactly
representable vlin evaluator), and with K = 1 slot in use it IS the sequential search."""
import numpy as np
import pytest

from oracle import evaluators as ev
from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import RefSearch, tree_dump

pytestmark = pytest.mark.gpu


def _check_tree(eng, game, env0, sims_expected):
    dump = eng.tree_dump(game)
    kids = {}
    for path in dump:
        if path:
            kids.setdefault(path[:-1], []).append(path)
    assert dump[()][0] == sims_expected
    collisions = 0
    for path, (n, w) in dump.items():
        env = env0.clone()
        for a in path:
            env.step(a)
        ended, winner = env.game_end_winner()
        n_kids = sum(dump[c][0] for c in kids.get(path, []))
        w_kids = sum(dump[c][1] for c in kids.get(path, []))
        here = n - n_kids  # simulations whose leaf was this node
        assert here >= 1 and n >= 1
        if ended:
            assert n_kids == 0
            v = 0.0 if winner == -1 else (1.0 if winner == env.current_player() else -1.0)
        else:
            v = ev.vlin_value(env.states, env.current_player())
        assert w + w_kids == -here * v, (path, n, w, here, v)  # exact: multiples of 1/8, no virtual loss left
        collisions += here - 1 if not ended else 0
    return collisions


@pytest.mark.parametrize('impl', ['level_sync', 'sequential'])
@pytest.mark.parametrize('K', [2, 5, 8, 16])
def test_visit_and_value_conservation(K, impl):
    from rlzero_amd.engine import MCTSEngine, SyntheticEvaluator, int_to_bits
    cases = [(6, 4, [], 203), (9, 5, [40, 41, 31], 160), (3, 3, [4, 0], 57)]
    for B, n, pre, sims in cases:
        envs = [RefGomoku.from_moves(B, n, pre), RefGomoku(B, n)]
        eng = MCTSEngine(B, n, n_games=2, n_playout=sims, sims_in_flight=K, in_flight_impl=impl, device='cuda:0')
        stones = np.array([[int_to_bits(e.bitboards()[0]), int_to_bits(e.bitboards()[1])] for e in envs], dtype=np.uint64)
        eng.set_roots(stones, [e.current_player() for e in envs], [e.last_move for e in envs], reset_trees=True)
        eng.simulate(SyntheticEvaluator('vlin'), sims)
        rn, _ = eng.root_stats()
        visits = eng.root_visits()
        eng.check()
        assert (rn == sims).all()
        for g, e in enumerate(envs):
            _check_tree(eng, g, e, sims)
            legal = e.leagel_actions()
            assert visits[g][legal].sum() <= sims - 1 and visits[g].sum() == visits[g][legal].sum()
        # tree reuse, then a second search on top of the kept subtree
        best = int(np.argmax(visits[0]))
        eng.advance([best, -2])
        eng.step([best, -1])
        envs[0].step(best)
        if not envs[0].game_end_winner()[0]:
            carried = int(eng.root_stats()[0][0])
            eng.simulate(SyntheticEvaluator('vlin'), sims)
            eng.check()
            assert int(eng.root_stats()[0][0]) == carried + sims
            _check_tree(eng, 0, envs[0], carried + sims)
        eng.close()


def _dump_bits(eng, g):
    return {p: (n, float(w).hex()) for p, (n, w) in eng.tree_dump(g).items()}


@pytest.mark.parametrize('mode', ['uct_ref', 'puct'])
def test_level_synchronous_kernel_equals_its_sequential_restatement(mode):
    """The production kernel of the mode (a workgroup of K waves per game, slots walked level by level) against the
    one-wave kernel that selects and backs up one slot after the other: identical trees, bit for bit -- with the
    synthetic evaluator and with the hand-written net (un-fused and fused routes), over several moves with tree reuse,
    Dirichlet noise on, ragged last steps (n_playout not a multiple of K)."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine, SyntheticEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(3)
    net = PolicyValueNet(9)
    with torch.no_grad():
        net.act_fc1.weight.mul_(10.0)
    net = net.to('cuda:0')
    for evaluator_kind, B, n, G, K, sims in (('vlin', 6, 4, 5, 7, 150), ('net', 9, 5, 6, 8, 123), ('vlin', 15, 5, 3, 16, 200),
                                             ('net', 9, 5, 4, 3, 64), ('vlin', (6, 7), 4, 4, 6, 90)):
        game = 'connect4' if isinstance(B, tuple) else 'gomoku'  # (Connect4: 7 column actions, gravity)
        engines = [MCTSEngine(B, n, n_games=G, n_playout=sims, sims_in_flight=K, in_flight_impl=impl, score_mode=mode,
                              add_noise=True, noise_seed=9, device='cuda:0', game=game) for impl in ('level_sync', 'sequential')]
        if evaluator_kind == 'net':
            net_b = net if B == 9 else None
            evaluators = [HipNetEvaluator(net_b, B, 'cuda:0', max_boards=G * K) for _ in engines]
        else:
            evaluators = [SyntheticEvaluator('vlin') for _ in engines]
        for eng in engines:
            eng.reset_games()
        for move in range(4):
            for eng, evl in zip(engines, evaluators):
                eng.simulate(evl, sims)
                eng.check()
            va, vb = engines[0].root_visits(), engines[1].root_visits()
            assert np.array_equal(va, vb), (evaluator_kind, move)
            assert np.array_equal(engines[0].root_wsum().view(np.uint64), engines[1].root_wsum().view(np.uint64))
            assert np.array_equal(engines[0].root_priors().view(np.uint32), engines[1].root_priors().view(np.uint32))
            for g in range(G):
                assert _dump_bits(engines[0], g) == _dump_bits(engines[1], g), (evaluator_kind, move, g)
            moves = [int(np.argmax(va[g])) for g in range(G)]
            for eng in engines:
                eng.advance(moves)
                _, ended = eng.step(moves)
            if ended.any():
                break
        for eng in engines:
            eng.close()


@pytest.mark.parametrize('impl', ['level_sync', 'sequential'])
def test_one_slot_in_use_is_the_sequential_search(impl):
    """sims_in_flight = 4 but every step told to use ONE slot: no other simulation is pending when a path is
    selected, so the tree must equal the oracle's (the virtual loss is put on and taken off exactly: multiples of 1/8)."""
    from rlzero_amd.engine import MCTSEngine, SyntheticEvaluator
    eng = MCTSEngine(6, 4, n_games=1, n_playout=150, sims_in_flight=4, in_flight_impl=impl, device='cuda:0')
    eng.reset_games()
    evaluator = SyntheticEvaluator('vlin')
    for _ in range(150):
        eng._in_flight(0, 1)
        eng.lib.rz_select_step(eng.handle, None, eng.stream())
        evaluator(eng)
        eng._in_flight(1, 0)
        eng.lib.rz_expand_backup(eng.handle, eng.logp.data_ptr(), eng.value.data_ptr(), eng.stream())
    eng.check()
    s = RefSearch(ev.vlin, 150, 5)
    s.simulate(RefGomoku(6, 4), 1.0)
    want = {p: (n, float(w).hex()) for p, (n, w) in tree_dump(s.root).items()}
    got = {p: (n, float(w).hex()) for p, (n, w) in eng.tree_dump(0).items()}
    assert got == want
    eng.close()


def test_batched_selfplay_with_the_net_and_graphs():
    """configs[1] geometry (9x9, 200 simulations, 64 games) with 8 simulations in flight through the production
    path: hand-written net on batches of 512 leaves, hipGraphs, noise, tree reuse; and the PUCT rule with K in flight."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(0)
    net = PolicyValueNet(9).to('cuda:0')
    for mode in ('uct_ref', 'puct'):
        sp = BatchedSelfPlay.for_network(net, board=9, n_in_row=5, n_games=64, n_playout=200, lanes=1, seed=3,
                                         sims_in_flight=8, score_mode=mode)
        assert sp.eng.n_leaves == 512
        sp._start(range(64), range(64))
        sp._set_active()
        for ply in range(4):
            carried = sp.eng.root_stats()[0].astype(np.int64) if ply else np.zeros(64, np.int64)
            sp._simulate()
            rn, _ = sp.eng.root_stats()
            visits = sp.eng.root_visits()
            assert (rn == carried + 200).all()
            if mode == 'uct_ref':
                assert (visits.sum(axis=1) <= rn - 1).all() and (visits.sum(axis=1) >= rn - 1 - 8 * 25).all()
            sp._simulate = lambda: None
            done = sp.play_move()
            del sp._simulate
            st = sp.eng.check()
            assert st.reuse_dropped == 0 and not done
        trajs = [(sp.slot_moves[s], sp.slot_pis[s]) for s in range(64)]
        assert all(len(m) == 4 and abs(np.sum(p[-1]) - 1.0) < 1e-9 for m, p in trajs)
        for lane in sp.lanes:
            lane.eng.close()


def test_whole_games_with_sims_in_flight():
    """BatchedSelfPlay.run() to the end of every game (slots refilled) with 4 simulations in flight: trajectories are
    complete and consistent, no arena flag, no dropped subtree."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(2)
    net = PolicyValueNet(6).to('cuda:0')
    sp = BatchedSelfPlay.for_network(net, board=6, n_in_row=4, n_games=16, n_playout=60, seed=4, sims_in_flight=4)
    trajs = sp.run(range(40))
    assert [t.game_id for t in trajs] == list(range(40))
    for t in trajs:
        assert 7 <= len(t.moves) <= 36 and len(set(t.moves)) == len(t.moves) and t.winner in (-1, 0, 1)
        assert abs(t.pis.sum(axis=1) - 1.0).max() < 1e-9
        w, data = t.as_reference_tuple()
        assert len(list(data)) == len(t.moves)
    for st in sp.check():
        assert st.reuse_dropped == 0
    for lane in sp.lanes:
        lane.eng.close()


def test_reference_api_player_with_sims_in_flight():
    """configs[0] (TicTacToe, 25 simulations) through AlphaZeroPlayer with 5 simulations in flight: whole games end,
    pi is a distribution over the legal moves and N(root) = 25 after a fresh search."""
    import torch
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts import AlphaZeroPlayer
    torch.manual_seed(1)
    np.random.seed(1)
    agent = AlphaZeroAgent(3, device='cuda:0')
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=25, c_puct=5, is_selfplay=True, sims_in_flight=5)
    env = GomokuEnv(3, 3)
    for _ in range(3):
        winner, data = GameControl(env).start_self_play(player, temperature=1.0)
        data = list(data)
        assert winner in (-1, 0, 1) and 5 <= len(data) <= 9
        for state, pi, z in data:
            assert abs(pi.sum() - 1.0) < 1e-9
    env.reset()
    player.reset_player()
    player.mcts.simulate(env, 1.0)
    assert int(player.mcts._engine.root_stats()[0][0]) == 25 and player.mcts._engine.n_leaves == 5
    player.mcts._engine.close()
