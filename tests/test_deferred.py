"""The deferred-priors route (include/rlzero_hip.h: rz_value_head; RZ_SCORE_UCT_REF, one simulation in flight, every board
size): a simulation step is trunk -> tree step (value head, backup, next selection); the policy half of
AlphaZeroAgent.policy_value_fn (alphazero_agent.py:41-45) and TreeNode.expand's priors (node.py:44-73) -- which the reference's
selection rule never reads (node.py:32-42,75-88) -- are written in one batch before anything reads them.

Pinned here: (1) the priors are, bit for bit, those of the route that writes them inside every tree step, the values agree with
it to f32 rounding and with the reference's CPU outputs to 1e-4; (2) the trees are EXACTLY the oracle's when it is fed the values
this route produces; (3) when the flushes happen -- store size, hipGraphs, tree reuse -- changes no bit."""
import numpy as np
import pytest

from oracle import evaluators as ev
from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import RefSearch, tree_dump

pytestmark = pytest.mark.gpu


def _positions(B, n, count, seed):
    rs = np.random.RandomState(seed)
    envs = [RefGomoku(B, n), RefGomoku.from_moves(B, n, [0])]
    while len(envs) < count:  # random non-terminal mid-game positions, up to a nearly full board
        e = RefGomoku(B, n)
        for m in rs.permutation(B * B)[:rs.randint(0, B * B - 1)]:
            e.step(int(m))
            if e.game_end_winner()[0]:
                break
        if not e.game_end_winner()[0]:
            envs.append(e)
    return envs


def _set_roots(eng, envs):
    from rlzero_amd.engine import int_to_bits
    stones = np.array([[int_to_bits(e.bitboards()[0]), int_to_bits(e.bitboards()[1])] for e in envs], dtype=np.uint64)
    eng.set_roots(stones, [e.current_player() for e in envs], [e.last_move for e in envs], reset_trees=True)


def _net(B, seed=None, g4=None):
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    net = PolicyValueNet(B)
    if g4 is not None:
        net.load_state_dict({k: torch.from_numpy(v) for k, v in ev.numpy_weights(B, int(g4['B%d_seed' % B])).items()})
    else:
        torch.manual_seed(seed)
        net = PolicyValueNet(B)
    return net


def _hex_tree(d):
    return {k: (n, float(w).hex()) for k, (n, w) in d.items()}


def _whole_tree(eng, game):
    """Every visited node of a game's tree, by its path of child ranks: (N, W bits, K, the bits of its K child priors) -- what
    the arena MEANS (reserved child slots that no visit has written yet hold whatever was there before)."""
    ar = eng.arena(game)
    out, stack = {}, [((), 0)]
    while stack:
        path, slot = stack.pop()
        k, pb = int(ar['K'][slot]), int(ar['PB'][slot])
        priors = ar['PRI'][pb:pb + k].view(np.uint32).tolist() if k > 0 else []
        out[path] = (int(ar['N'][slot]), float(ar['W'][slot]).hex(), k, tuple(priors))
        if k > 0:
            assert len(priors) == k and all(np.isfinite(ar['PRI'][pb:pb + k])) and (ar['PRI'][pb:pb + k] > 0).all()
            for r in range(int(ar['NV'][slot])):
                stack.append((path + (r, ), int(ar['FC'][slot]) + r))
    return out


def test_deferred_priors_are_the_in_step_route_s_bits(g4):
    """One expansion of 16 positions per board size on both routes: the priors (read after the flush) are the same bits -- the
    policy GEMM over the store is k_heads_split on the same f16 pieces, the softmax and the noise the same operations --, the
    values agree to f32 rounding (the value head's first layer is summed by the game's workgroup instead of the GEMM), and at
    15 x 15 both agree with the reference's policy_value_fn output (1e-4).  Then three more simulations: same roots."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    for B, n, noise in ((15, 5, False), (15, 5, True), (11, 5, True), (13, 5, False), (16, 5, True), (3, 3, True), (6, 4, False), (9, 5, True), (10, 5, False)):
        net = _net(B, seed=B, g4=g4 if B == 15 else None)
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=16)
        envs = _positions(B, n, 16, seed=B)
        got = {}
        for deferred in (True, False):
            evaluator.deferred_priors = deferred
            eng = MCTSEngine(B, n, n_games=len(envs), n_playout=8, device='cuda:0', add_noise=noise, noise_seed=3)
            assert evaluator.deferred_ok(eng) == deferred
            _set_roots(eng, envs)
            eng.sim_chunk(evaluator, 1)  # the first simulation expands every root
            assert (eng._def_pending == 1) == deferred
            pri = eng.root_priors().copy()  # (flushes)
            assert eng._def_pending == 0
            rn, rw = eng.root_stats()
            assert (rn == 1).all()
            eng.sim_chunk(evaluator, 3)
            pri4 = eng.root_priors().copy()
            rn4, _ = eng.root_stats()
            eng.check()
            got[deferred] = (pri, rw.copy(), pri4, rn4.copy())
            eng.close()
        assert np.array_equal(got[True][0].view(np.uint32), got[False][0].view(np.uint32)), B
        assert np.array_equal(got[True][2].view(np.uint32), got[False][2].view(np.uint32)), B
        assert np.max(np.abs(got[True][1] - got[False][1])) <= 2e-6 and (got[True][3] == 4).all()
        for g, e in enumerate(envs):  # priors exist for the legal moves only
            illegal = sorted(set(range(B * B)) - set(e.leagel_actions()))
            assert not got[True][0][g][illegal].any() and 0.0 < got[True][0][g].astype(np.float64).sum() <= 1.0 + 1e-5
        if B == 15 and not noise:
            acts = [int(a) for a in g4['B15_pvf_acts']]
            want_p, want_v = g4['B15_pvf_probs'].astype(np.float64), float(g4['B15_pvf_value'])
            assert np.max(np.abs(got[True][0][1][acts].astype(np.float64) - want_p)) <= 1e-4
            assert abs(-got[True][1][1] - want_v) <= 1e-4
        evaluator.hip.close()


def test_deferred_search_is_the_oracle_s_search_on_this_route_s_values():
    """Whole searches at 15 x 15 on the deferred route == the oracle's sequential search (node.py, alphazero_mcts.py:42-94) fed,
    leaf by leaf, the value this route gives that position (a one-game engine expanding the position as its root: the value of a
    board does not depend on its batch): every node's N and W, bit for bit -- with Dirichlet noise on (it touches priors only)."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n, sims = 15, 5, 150
    net = _net(B, seed=1)
    evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=8)
    probe_eval = HipNetEvaluator(net, B, 'cuda:0', max_boards=1)
    probe = MCTSEngine(B, n, n_games=1, n_playout=4, device='cuda:0')
    cache = {}

    def pvf(env):
        key = (env.bitboards(), env.current_player(), env.last_move)
        if key not in cache:
            _set_roots(probe, [env])
            probe.sim_chunk(probe_eval, 1)
            cache[key] = -float(probe.root_stats()[1][0])
        legal = env.leagel_actions()
        return [(a, 1.0 / len(legal)) for a in legal], cache[key]

    envs = [RefGomoku(B, n), RefGomoku.from_moves(B, n, [112, 113, 97]),
            RefGomoku.from_moves(B, n, [112, 111, 113, 110, 114, 109, 115]),   # four in a row: wins and terminal leaves nearby
            _positions(B, n, 6, seed=4)[5]]
    eng = MCTSEngine(B, n, n_games=len(envs), n_playout=sims, device='cuda:0', add_noise=True, noise_seed=9)
    _set_roots(eng, envs)
    eng.simulate(evaluator, sims)
    eng.check()
    assert eng._def_pending == sims
    for g, env in enumerate(envs):
        ref = RefSearch(pvf, sims, 5)
        ref.simulate(env, 1.0)
        assert _hex_tree(eng.tree_dump(g)) == _hex_tree(tree_dump(ref.root)), 'game %d' % g
    for e_ in (eng, probe):
        e_.close()
    evaluator.hip.close()
    probe_eval.hip.close()


def test_when_the_flushes_happen_changes_no_bit():
    """The same searches with tree reuse over three moves: (a) one flush per move, eager launches; (b) a store of 16 slots
    (a flush every 16 steps, also in the middle of a chunk); (c) hipGraphs of 16 steps.  Every record of every arena -- N, W,
    the priors of every expanded node -- is the same."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n, sims = 13, 5, 90
    net = _net(B, seed=2)
    envs = _positions(B, n, 6, seed=8)
    dumps = {}
    for mode in ('move', 'small_store', 'graph'):
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=len(envs))
        evaluator.resident_search = False   # (the two-launch step and its hipGraphs; the resident search has a test of its own)
        eng = MCTSEngine(B, n, n_games=len(envs), n_playout=sims, device='cuda:0', add_noise=True, noise_seed=5)
        if mode == 'small_store':
            eng.deferred_max_bytes = 1   # -> the minimum of 16 slots
        if mode == 'graph':
            eng.reset_games()
            eng.warm_graph(evaluator, 16)
        _set_roots(eng, envs)
        eng.set_noise_keys()   # (the games' noise streams restart: the graph warm-up has drawn from them)
        record = []
        for move in range(3):
            eng.simulate(evaluator, sims, use_graph=mode == 'graph', sims_per_graph=16)
            if mode == 'small_store':
                assert eng._def_slots == 16 and 0 < eng._def_pending <= 16
            else:
                assert eng._def_pending == sims
            visits = eng.root_visits()
            assert eng._def_pending == 0 and (visits.sum(axis=1) == eng.root_stats()[0] - 1).all()
            record.append(visits.copy())
            record.append([_whole_tree(eng, g) for g in range(len(envs))])
            eng.advance(visits.argmax(axis=1).astype(np.int32))   # keep the most visited child's subtree
            eng.step(visits.argmax(axis=1).astype(np.int32))
        st = eng.check()
        assert st.reuse_dropped == 0
        dumps[mode] = record
        eng.close()
        evaluator.hip.close()
    for mode in ('small_store', 'graph'):
        for a, b in zip(dumps['move'], dumps[mode]):
            if isinstance(a, np.ndarray):
                assert np.array_equal(a, b), mode
            else:
                assert a == b, mode


def test_an_evaluator_of_another_route_takes_over_cleanly():
    """Pending priors are written before another evaluator searches the same trees (sim_chunk flushes), before the roots are set
    anew, and before the weights of the deferring evaluator change."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine, SyntheticEvaluator
    B, n = 15, 5
    net = _net(B, seed=3).to('cuda:0')
    evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=4)
    eng = MCTSEngine(B, n, n_games=4, n_playout=40, device='cuda:0')
    eng.reset_games()
    eng.sim_chunk(evaluator, 10)
    assert eng._def_pending == 10
    eng.sim_chunk(SyntheticEvaluator('vlin'), 5)
    assert eng._def_pending == 0 and (eng.root_stats()[0] == 15).all()
    pri = eng.arena(0)['PRI']
    assert np.isfinite(pri).all() and (pri > 0).all()   # every reserved block was written (by the flush or by the in-step route)
    eng.sim_chunk(evaluator, 10)
    assert eng._def_pending == 10
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.01)
    evaluator.refresh_if_changed()
    assert eng._def_pending == 0
    eng.sim_chunk(evaluator, 3)
    eng.reset_games()
    assert eng._def_pending == 0
    eng.check()
    eng.close()
    evaluator.hip.close()


def test_resident_search_is_the_two_launch_step_in_one_launch():
    """rz_net_search_resident: for a batch of at most one game per CU the simulations of a search run as ONE launch, one workgroup per
    game -- trunk, value head, expand / backup and the next selection back to back, the leaf handed over through LDS.  It is the
    deferred route's arithmetic and bookkeeping: over three moves with tree reuse every visited node's N, W and priors equal those
    of the two-launch step, bit for bit; both trunk kernels (boards of 3 .. 16 rows), noise, an idle slot, games that end inside the three moves."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    for (rows, cols), sims, noise in (((15, 15), 90, True), ((11, 11), 60, False), ((16, 16), 40, True), ((13, 13), 50, True),
                                     ((9, 9), 80, True), ((6, 6), 70, False), ((3, 3), 25, True), ((10, 10), 40, True), ((8, 8), 40, False)):
        B = rows
        torch.manual_seed(rows)
        net = _net(B, seed=rows)
        n_row = 5 if B >= 8 else (4 if B == 6 else 3)
        envs = _positions(B, n_row, 7, seed=rows)
        dumps = {}
        for resident in (True, False):
            evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=len(envs))
            evaluator.resident_search = resident
            eng = MCTSEngine(B, n_row, n_games=len(envs), n_playout=sims, device='cuda:0', add_noise=noise, noise_seed=5)
            assert evaluator.resident_ok(eng) == resident and evaluator.deferred_ok(eng)
            _set_roots(eng, envs)
            eng.set_noise_keys()
            active = np.ones(len(envs), dtype=np.uint8)
            active[3] = 0   # an idle slot: its workgroup ends at once, its tree stays a fresh root
            eng.set_active(active)
            record = []
            for move in range(3):
                eng.simulate(evaluator, sims, use_graph=False)
                assert eng._def_pending == sims
                visits = eng.root_visits()
                record.append(visits.copy())
                record.append([_whole_tree(eng, g) for g in range(len(envs))])
                playing = (active > 0) & (visits.sum(axis=1) > 0)   # (a root that is terminal already has no visited child)
                moves = np.where(playing, visits.argmax(axis=1), -2).astype(np.int32)
                eng.advance(moves)
                _, ended = eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
                active = (playing & (np.asarray(ended) == 0)).astype(np.uint8)   # games that ended stop searching
                eng.set_active(active)
            st = eng.check()
            assert st.reuse_dropped == 0
            dumps[resident] = record
            eng.close()
            evaluator.hip.close()
        for a, b in zip(dumps[True], dumps[False]):
            if isinstance(a, np.ndarray):
                assert np.array_equal(a, b), (rows, cols)
                assert (a[3] == 0).all()
            else:
                assert a == b, (rows, cols)


def test_resident_search_beyond_two_games_per_cu_runs_in_rounds():
    """k_delta_res takes a grid beyond the two workgroups a CU holds (rz_net_search_resident: any number of games with the base cache
    reserved): the dispatcher hands a CU's free half to the next game as a search ends, and no game sees another -- 800 games (a full
    round of 512 and a partial one on 256 CUs) over two moves with tree reuse leave the root visits of the two-launch step on every
    game, and its whole trees on a sample of them, bit for bit."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n_row, sims, G = 12, 5, 24, 800
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert G > 3 * n_cus or n_cus != 256
    net = _net(B, seed=21)
    envs = _positions(B, n_row, G, seed=4)
    sample = list(range(0, G, 37)) + [G - 1]
    dumps = {}
    for resident in (True, False):
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=G)
        evaluator.resident_search = resident
        eng = MCTSEngine(B, n_row, n_games=G, n_playout=sims, device='cuda:0', add_noise=True, noise_seed=9)
        assert evaluator.resident_ok(eng) == resident and evaluator.deferred_ok(eng) and evaluator.resident_delta_ok(eng)
        _set_roots(eng, envs)
        eng.set_noise_keys()
        record = []
        for move in range(2):
            eng.simulate(evaluator, sims, use_graph=False)
            visits = eng.root_visits()
            record.append(visits.copy())
            record.append([_whole_tree(eng, g) for g in sample])
            playing = visits.sum(axis=1) > 0
            moves = np.where(playing, visits.argmax(axis=1), -2).astype(np.int32)
            eng.advance(moves)
            _, ended = eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
            eng.set_active((playing & (np.asarray(ended) == 0)).astype(np.uint8))
        if resident:
            st = evaluator.hip.delta_stats()
            assert st['delta'] > 10 * st['no_base'] > -1   # (k_delta_res ran, its leaves against their roots' bases: a rare deep one takes the passes without)
        assert eng.check().reuse_dropped == 0
        dumps[resident] = record
        eng.close()
        evaluator.hip.close()
    for a, b in zip(dumps[True], dumps[False]):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


@pytest.mark.parametrize('B,n_row,G', [(6, 4, 300), (3, 3, 300), (7, 5, 560)])
def test_resident_search_on_the_compact_grid(B, n_row, G):
    """Boards of up to 7 columns: from half a chip of games on the resident search runs on k_trunk_split's COMPACT LDS grid (69 KB: two
    games per CU, any number of games per launch -- 560 games are a full round and a partial one).  The grid changes where a position
    lives in LDS, not one product: root visits on every game and whole trees on a sample equal the two-launch step's bit for bit over two
    moves with tree reuse (300 games of 6x6: two N-tiles; 3x3: one; 7x7: two, in rounds)."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    sims = 30
    net = _net(B, seed=B)
    base = _positions(B, n_row, 40, seed=B + 1)
    envs = [base[i % len(base)] for i in range(G)]
    sample = list(range(0, G, 29)) + [G - 1]
    dumps = {}
    for resident in (True, False):
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=G)
        evaluator.resident_search = resident
        eng = MCTSEngine(B, n_row, n_games=G, n_playout=sims, device='cuda:0', add_noise=True, noise_seed=3)
        assert evaluator.hip.compact_resident() and evaluator.resident_per_cu(eng) == 2
        assert evaluator.resident_ok(eng) == resident and evaluator.deferred_ok(eng) and 2 * G > n_cus
        _set_roots(eng, envs)
        eng.set_noise_keys()   # (a noise stream per game: games on the same root still search differently)
        record = []
        for move in range(2):
            eng.simulate(evaluator, sims, use_graph=False)
            visits = eng.root_visits()
            record.append(visits.copy())
            record.append([_whole_tree(eng, g) for g in sample])
            playing = visits.sum(axis=1) > 0
            moves = np.where(playing, visits.argmax(axis=1), -2).astype(np.int32)
            eng.advance(moves)
            _, ended = eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
            eng.set_active((playing & (np.asarray(ended) == 0)).astype(np.uint8))
        assert eng.check().reuse_dropped == 0
        dumps[resident] = record
        eng.close()
        evaluator.hip.close()
    for a, b in zip(dumps[True], dumps[False]):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


@pytest.mark.parametrize('score_mode', ['puct', 'uct_ref'])
def test_three_launch_step_on_the_receptive_field_trunk(score_mode):
    """The three-launch step (trunk -> FC GEMM -> tree step: the opt-in PUCT rule, and UCT_REF with deferred_priors = False) on boards of
    11 .. 16 rows evaluates its leaves by receptive fields too (rz_net_delta_trunk_engine: the FC GEMM's own f16 tiles, policy and value
    K-steps): whole trees -- N, W bits, priors bits -- equal the full-board trunk's over three moves with tree reuse, noise, an idle
    slot; and the counters say the kernel ran against bases."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    for B, sims in ((15, 70), (12, 50)):
        net = _net(B, seed=B + 40)
        envs = _positions(B, 5, 9, seed=B)
        dumps = {}
        for delta in (True, False):
            evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=len(envs))
            evaluator.delta_trunk = delta
            evaluator.deferred_priors = False
            eng = MCTSEngine(B, 5, n_games=len(envs), n_playout=sims, device='cuda:0', add_noise=True, noise_seed=6, score_mode=score_mode)
            assert not evaluator.deferred_ok(eng) and evaluator.delta_three_launch_ok(eng) == delta
            _set_roots(eng, envs)
            eng.set_noise_keys()
            active = np.ones(len(envs), dtype=np.uint8)
            active[2] = 0
            eng.set_active(active)
            record = []
            for move in range(3):
                eng.simulate(evaluator, sims, use_graph=False)   # (hipGraph replays of this step: the full-size layout test's PUCT case)
                visits = eng.root_visits()
                record.append(visits.copy())
                record.append([_whole_tree(eng, g) for g in range(len(envs))])
                playing = (active > 0) & (visits.sum(axis=1) > 0)
                moves = np.where(playing, visits.argmax(axis=1), -2).astype(np.int32)
                eng.advance(moves)
                _, ended = eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
                active = (playing & (np.asarray(ended) == 0)).astype(np.uint8)
                eng.set_active(active)
            eng.check()
            if delta:
                st = evaluator.hip.delta_stats()
                assert st['delta'] > 10 * st['no_base'] and st['delta'] > sims
            dumps[delta] = record
            eng.close()
            evaluator.hip.close()
        for a, b in zip(dumps[True], dumps[False]):
            assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b, (B, score_mode)


def test_two_engines_sharing_one_evaluator():
    """An evaluator's feature store holds the pending leaves of one engine at a time: when a second engine searches with the same
    evaluator the first one's priors are written first, so interleaved searches leave the trees they would leave alone."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n = 15, 5
    net = _net(B, seed=6)
    envs = _positions(B, n, 5, seed=11)
    trees = {}
    for shared in (True, False):
        ev_a = HipNetEvaluator(net, B, 'cuda:0', max_boards=8)
        ev_b = ev_a if shared else HipNetEvaluator(net, B, 'cuda:0', max_boards=8)
        a = MCTSEngine(B, n, n_games=len(envs), n_playout=60, device='cuda:0', add_noise=True, noise_seed=2)
        b = MCTSEngine(B, n, n_games=3, n_playout=60, device='cuda:0', add_noise=True, noise_seed=3)
        _set_roots(a, envs)
        _set_roots(b, envs[:3])
        a.sim_chunk(ev_a, 25)
        b.sim_chunk(ev_b, 30)          # (shared: a's 25 pending steps are flushed before b's leaves take the store's slots)
        assert a._def_pending == (0 if shared else 25) and b._def_pending == 30
        a.sim_chunk(ev_a, 35)
        assert b._def_pending == (0 if shared else 30)
        trees[shared] = ([_whole_tree(a, g) for g in range(len(envs))], [_whole_tree(b, g) for g in range(3)])
        a.close()
        b.close()
    assert trees[True] == trees[False]


def test_a_store_that_moves_drops_the_graphs_captured_on_it():
    """hipGraphs hold device addresses by value.  Engine A (5 games, 60 simulations) captures its steps on a shared evaluator; engine
    B (3 games, 140 simulations) then needs more store slots: rz_net_deferred_reserve grows the store -- in BOTH directions, A's five
    boards keep fitting -- and A's graphs are dropped, not replayed into freed memory: simulate(use_graph=True) says so, a new
    warm_graph captures against the new store, and the trees are those of an engine that never shared anything."""
    from rlzero_amd._hip import HipError
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n = 13, 5
    net = _net(B, seed=12)
    envs = _positions(B, n, 5, seed=3)
    shared = HipNetEvaluator(net, B, 'cuda:0', max_boards=8)
    shared.resident_search = False
    a = MCTSEngine(B, n, n_games=len(envs), n_playout=60, device='cuda:0', add_noise=True, noise_seed=2)
    b = MCTSEngine(B, n, n_games=3, n_playout=140, device='cuda:0', add_noise=True, noise_seed=3)
    a.reset_games()
    a.warm_graph(shared, 16)
    assert len(a._graphs) == 1
    _set_roots(a, envs)
    a.set_noise_keys()
    a.simulate(shared, 32, use_graph=True, sims_per_graph=16)
    _set_roots(b, envs[:3])
    b.sim_chunk(shared, 140)   # (A's pending leaves are flushed, then the store grows to 140 slots)
    assert a._def_pending == 0 and b._def_pending == 140 and not a._graphs
    with pytest.raises(HipError, match='graphs were dropped'):
        a.simulate(shared, 16, use_graph=True, sims_per_graph=16)
    a.flush_deferred()
    b.flush_deferred()
    a.warm_graph(shared, 16)   # (captures nothing into the trees: the warm-up's simulations are part of what is compared below)
    a.simulate(shared, 16, use_graph=True, sims_per_graph=16)
    got = [_whole_tree(a, g) for g in range(len(envs))]
    # the same sequence on an evaluator of its own: 32 simulations, then the 3 of the second warm-up, then 16
    own = HipNetEvaluator(net, B, 'cuda:0', max_boards=8)
    own.resident_search = False
    c = MCTSEngine(B, n, n_games=len(envs), n_playout=60, device='cuda:0', add_noise=True, noise_seed=2)
    _set_roots(c, envs)
    c.set_noise_keys()
    c.sim_chunk(own, 32 + 3 + 16)
    assert got == [_whole_tree(c, g) for g in range(len(envs))]
    for e_ in (a, b, c):
        e_.check()
        e_.close()
    shared.hip.close()
    own.hip.close()


def test_a_flush_from_another_stream_waits_for_the_pending_steps():
    """The pending steps of a search live on the stream they were enqueued on; a read-out from another stream (here the default
    stream, while the lane's stream still holds the search) flushes behind them: the priors are those of a search read on its own stream."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n = 15, 5
    net = _net(B, seed=13)
    envs = _positions(B, n, 6, seed=5)
    out = []
    for other_stream in (True, False):
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=len(envs))
        evaluator.resident_search = False
        eng = MCTSEngine(B, n, n_games=len(envs), n_playout=200, device='cuda:0', add_noise=True, noise_seed=7)
        _set_roots(eng, envs)
        eng.set_noise_keys()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eng.sim_chunk(evaluator, 200)   # ~10 ms of launches: still running when the read-out below is enqueued
            if not other_stream:
                pri = eng.root_priors().copy()
        if other_stream:
            pri = eng.root_priors().copy()   # (current stream = the default stream)
        torch.cuda.synchronize()
        out.append((pri, [_whole_tree(eng, g) for g in range(len(envs))]))
        eng.check()
        eng.close()
        evaluator.hip.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and out[0][1] == out[1][1]


def test_a_value_head_narrower_than_the_board_is_refused():
    """The tree step sizes its bitboard arithmetic by the value head's width (16 .. 128 groups of four inputs hold the 2 S inputs of
    boards of up to 32 .. 256 cells: a TicTacToe or Connect4 board is one 64-bit word of a colour, rz_tree.h): a head that cannot
    hold this engine's board is an error at the C ABI, not a search on half a board."""
    import ctypes
    from rlzero_amd import _hip
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, n = 9, 5
    net = _net(B, seed=3)
    evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=2)
    evaluator.resident_search = False
    eng = MCTSEngine(B, n, n_games=2, n_playout=20, device='cuda:0')
    eng.reset_games()
    eng.sim_chunk(evaluator, 4)   # (the route works; a head as the trunk leaves it:)
    head = evaluator.hip.trunk_leaves_deferred(eng)
    assert head.groups == 64 and head.ld == 256   # 2 x 81 inputs
    narrow = _hip.RzValueHead()
    ctypes.memmove(ctypes.byref(narrow), ctypes.byref(head), ctypes.sizeof(head))
    narrow.groups, narrow.ld = 16, 64
    rc = eng.lib.rz_tree_step_deferred(eng.handle, ctypes.byref(narrow), None)
    assert rc != 0 and b'do not hold' in eng.lib.rz_last_error()
    eng.close()
    evaluator.hip.close()
