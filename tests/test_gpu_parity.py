"""Parity of the HIP path (through the C ABI) with the oracle and the golden fixtures.

Bit-exact: visit counts, W (fp64 bit patterns), moves, winners, z, observation planes.
pi to 1e-12 (same numpy expression on identical integers).  Net outputs within the 1e-4
fp32 tolerance BASELINE.json states.  Run on the GPU box: ``pytest -m gpu``.
"""
import hashlib
import math

import numpy as np
import pytest
from conftest import bits_of_planes, unhex

from oracle import evaluators as ev
from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import (RefPlayer, RefSearch, inverse_cdf_choice, self_play_game, tree_dump)

pytestmark = pytest.mark.gpu

EVALS = {'v0': ev.v0, 'vlin': ev.vlin}


def _engine(*args, **kw):
    from rlzero_amd.engine import MCTSEngine
    return MCTSEngine(*args, **kw)


def _stones_of(envs):
    from rlzero_amd.engine import int_to_bits
    return np.array([[int_to_bits(e.bitboards()[0]), int_to_bits(e.bitboards()[1])] for e in envs],
                    dtype=np.uint64)


def _set_roots(eng, envs, **kw):
    eng.set_roots(_stones_of(envs), [e.current_player() for e in envs], [e.last_move for e in envs], **kw)


def _hex_tree(dump):
    return {p: (n, float(w).hex()) for p, (n, w) in dump.items()}


# ------------------------------------------------------------------ arithmetic
def test_uct_arithmetic_bit_exact():
    """q + c*u on the device == CPython's two-rounding fp64 expression (node.py:83-87)."""
    eng = _engine(15, 5, n_games=1, n_playout=800)
    rs = np.random.RandomState(0)
    count = 200000
    n = rs.randint(1, 4000, size=count).astype(np.int32)
    npar = np.maximum(n, rs.randint(1, eng.log_table_size() - 1, size=count)).astype(np.int32)
    w = np.where(rs.rand(count) < 0.5, rs.randint(-8 * 4000, 8 * 4000, size=count) / 8.0,
                 rs.standard_normal(count) * n)
    w = np.clip(w, -n.astype(np.float64), n.astype(np.float64))
    for c in (5.0, 0.5, 1.25, 0.0):
        got = eng.uct_scores(w, n, npar, c)
        want = np.array([wi / ni + c * math.sqrt(math.log(pi) / ni)
                         for wi, ni, pi in zip(w.tolist(), n.tolist(), npar.tolist())])
        assert got.view(np.uint64).tolist() == want.view(np.uint64).tolist()
    # the table really is the host libm's log: re-uploading math.log changes nothing
    base = eng.uct_scores(w, n, npar, 5.0)
    eng.upload_log_table([0.0] + [math.log(i) for i in range(1, eng.log_table_size())])
    assert (eng.uct_scores(w, n, npar, 5.0).view(np.uint64) == base.view(np.uint64)).all()
    # unvisited child / parent -> +inf
    assert np.isinf(eng.uct_scores([0.0, 1.0], [0, 3], [5, 0], 5.0)).all()
    eng.close()


# ------------------------------------------------------------------ rules
@pytest.mark.parametrize('group', ['random', 'handmade'])
def test_rules_on_device(g1, group):
    """rz_step_games / rz_encode_root_obs vs every ply of the rules fixtures."""
    by_board = {}
    for case in g1[group]:
        by_board.setdefault((case['B'], case['n']), []).append(case)
    for (B, n), cases in by_board.items():
        eng = _engine(B, n, n_games=len(cases), n_playout=4)
        eng.reset_games()
        for ply_i in range(max(len(c['plies']) for c in cases)):
            moves = np.array([c['plies'][ply_i]['a'] if ply_i < len(c['plies']) else -1 for c in cases],
                             dtype=np.int32)
            winner, ended = eng.step(moves)
            obs = eng.root_obs().cpu().numpy()
            stones, to_move, last = eng.get_roots()
            for g, c in enumerate(cases):
                if ply_i >= len(c['plies']):
                    continue
                ply = c['plies'][ply_i]
                assert bool(ended[g]) == ply['ended'] and int(winner[g]) == ply['winner'], (c.get('name'), ply_i)
                assert bits_of_planes(obs[g]) == ply['obs']
                assert int(to_move[g]) == ply['to_move'] and int(last[g]) == ply['last']
        eng.check()
        eng.close()


def test_illegal_move_is_flagged():
    from rlzero_amd._hip import HipError
    eng = _engine(3, 3, n_games=2, n_playout=4)
    eng.reset_games()
    eng.step([4, 4])
    eng.step([4, 0])
    with pytest.raises(HipError):
        eng.check()
    eng.close()


# ------------------------------------------------------------------ search (G2)
def _check_case(eng, evaluator, rec, env):
    from rlzero_amd.selfplay import visits_to_pi
    eng.simulate(evaluator, rec['n_playout'])
    visits, wsum = eng.root_visits()[0], eng.root_wsum()[0]
    rn, rw = eng.root_stats()
    eng.check()
    assert int(rn[0]) == rec['root_N'] and float(rw[0]).hex() == rec['root_W']
    acts = rec['acts']
    assert env.leagel_actions() == acts
    assert [int(visits[a]) for a in acts] == rec['N']
    # unvisited children have W == int 0 in the reference -> '0x0.0p+0'
    assert [float(wsum[a]).hex() for a in acts] == rec['W']
    occupied = [a for a in range(eng.n_cells) if a not in acts]
    assert not visits[occupied].any()
    pi = visits_to_pi(visits[acts], rec['T'])
    assert np.max(np.abs(pi - np.array([unhex(p) for p in rec['pi']]))) <= 1e-12
    dump = _hex_tree(eng.tree_dump(0))
    assert len(dump) == rec['n_nodes']
    if 'tree' in rec:
        assert dump == {tuple(p): (n, w) for p, n, w in rec['tree']}
    else:
        h = hashlib.sha1()
        for path in sorted(dump, key=list):
            h.update(repr((path, dump[path][0], dump[path][1])).encode())
        assert h.hexdigest() == rec['tree_sha1']


def test_search_synthetic_golden(g2):
    from rlzero_amd.engine import SyntheticEvaluator
    for rec in g2['cases']:
        env = RefGomoku.from_moves(rec['B'], rec['n'], rec['pre'])
        eng = _engine(rec['B'], rec['n'], n_games=1, n_playout=rec['n_playout'], c_puct=rec['c_puct'])
        _set_roots(eng, [env], reset_trees=True)
        _check_case(eng, SyntheticEvaluator(rec['eval']), rec, env)
        eng.close()


def test_search_host_callable_golden(g2):
    """Same fixtures through the generic evaluator plug-in (a Python callable per leaf)."""
    from rlzero_amd.engine import HostEvaluator
    from rlzero_amd.games import GomokuEnv
    for rec in g2['cases']:
        if rec['n_playout'] > 400:
            continue
        B, n = rec['B'], rec['n']
        env = RefGomoku.from_moves(B, n, rec['pre'])
        eng = _engine(B, n, n_games=1, n_playout=rec['n_playout'], c_puct=rec['c_puct'])
        _set_roots(eng, [env], reset_trees=True)
        host = HostEvaluator(EVALS[rec['eval']],
                             lambda s0, s1, tm, last, B=B, n=n: GomokuEnv.from_bitboards(B, n, s0, s1, tm, last))
        _check_case(eng, host, rec, env)
        eng.close()


def test_search_batch_of_positions_vs_oracle():
    """64 different 9x9 positions searched in lock-step == 64 independent oracle searches
    (BASELINE config 2: 9x9, 200 sims, 64 games)."""
    from rlzero_amd.engine import SyntheticEvaluator
    rs = np.random.RandomState(99)
    envs = []
    while len(envs) < 64:
        k = rs.randint(0, 40)
        env = RefGomoku.from_moves(9, 5, [])
        for m in rs.permutation(81)[:k]:
            env.step(int(m))
            if env.game_end_winner()[0]:
                break
        if not env.game_end_winner()[0]:
            envs.append(env)
    eng = _engine(9, 5, n_games=64, n_playout=200)
    _set_roots(eng, envs, reset_trees=True)
    eng.simulate(SyntheticEvaluator('vlin'), 200)
    visits, wsum = eng.root_visits(), eng.root_wsum()
    eng.check()
    for g, env in enumerate(envs):
        s = RefSearch(ev.vlin, 200, 5)
        acts, _ = s.simulate(env, 1.0)
        assert [int(visits[g, a]) for a in acts] == [k.n for k in s.root.kids]
        assert [float(wsum[g, a]).hex() for a in acts] == [float(k.w).hex() for k in s.root.kids]
        if g % 8 == 0:
            assert _hex_tree(eng.tree_dump(g)) == _hex_tree(tree_dump(s.root))
    eng.close()


def test_edge_cases_max_board_single_game_near_full_board_empty_batch():
    """Edges of the domain: the largest board the ABI admits (16x16, 4 bitboard words all in use), one game
    per engine, a board with a single empty cell (every simulation ends in a terminal leaf or the last
    move), a search longer than a child vector's first capacities (growth 4 -> 8 -> ... -> K), and the
    evaluator on an empty batch."""
    import torch
    from rlzero_amd.engine import HipNet, SyntheticEvaluator
    # 16 x 16, n = 5, one game: 600 simulations from the empty board visit all 256 children (vector grown to K)
    eng = _engine(16, 5, n_games=1, n_playout=600)
    eng.reset_games()
    eng.simulate(SyntheticEvaluator('vlin'), 600)
    eng.check()
    s = RefSearch(ev.vlin, 600, 5)
    s.simulate(RefGomoku(16, 5), 1.0)
    assert _hex_tree(eng.tree_dump(0)) == _hex_tree(tree_dump(s.root))
    assert int(eng.root_visits()[0].sum()) == 599 and (eng.root_visits()[0] > 0).all()  # N(root) = 1 + sum
    eng.close()
    # a 6x6 board (n = 4) filled without a winner except for the last 1 / 2 cells: colour = (column pair + row)
    # parity gives runs of at most 2 in every direction
    cells = [r * 6 + c for r in range(6) for c in range(6)]
    a_cells = [c for c in cells if ((c % 6) // 2 + c // 6) % 2 == 0]
    b_cells = [c for c in cells if ((c % 6) // 2 + c // 6) % 2 == 1]
    for keep in (1, 2):
        seq = []
        for x, y in zip(a_cells, b_cells):
            seq += [x, y]
        seq = seq[:36 - keep]
        env = RefGomoku.from_moves(6, 4, seq)
        assert not env.game_end_winner()[0] and len(env.leagel_actions()) == keep
        eng = _engine(6, 4, n_games=2, n_playout=40)
        _set_roots(eng, [env, env], reset_trees=True)
        eng.simulate(SyntheticEvaluator('vlin'), 40)
        eng.check()
        ref = RefSearch(ev.vlin, 40, 5)
        ref.simulate(env, 1.0)
        for g in (0, 1):
            assert _hex_tree(eng.tree_dump(g)) == _hex_tree(tree_dump(ref.root))
        eng.close()
    # evaluator: empty batch is a no-op, a 1-board batch and a ragged 33-board batch agree row by row
    torch.manual_seed(4)
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    net = PolicyValueNet(16)
    hip = HipNet(16, 'cuda:0', max_boards=64).load_state_dict(net.state_dict())
    x = (torch.rand((33, 4, 16, 16), device='cuda:0') > 0.5).float()
    lp0, v0 = hip.forward(x[:0])
    assert lp0.shape[0] == 0 and v0.shape[0] == 0
    lp, v = hip.forward(x)
    lp1, v1 = hip.forward(x[7:8])
    assert torch.equal(lp1[0], lp[7]) and torch.equal(v1[0], v[7])
    with torch.no_grad():
        want_lp, want_v = net(x.cpu())
    assert float((lp.cpu() - want_lp).abs().max()) <= 1e-4 and float((v.cpu() - want_v[:, 0]).abs().max()) <= 1e-4
    hip.close()


def test_full_size_known_answer_and_invariants():
    """15x15, 800 sims (the metric's configuration): SURVEY.md Appendix B KAT 6 + the
    size-independent invariants N(root) = n_playout, N(node) = 1 + sum N(children),
    visited children form a prefix."""
    from rlzero_amd.engine import SyntheticEvaluator
    eng = _engine(15, 5, n_games=8, n_playout=800)
    eng.reset_games()
    eng.simulate(SyntheticEvaluator('vlin'), 800)
    visits = eng.root_visits()
    rn, rw = eng.root_stats()
    eng.check()
    assert (rn == 800).all() and (rw == -0.5).all()
    for g in (0, 7):
        assert hashlib.sha1(visits[g].astype(np.int32).tobytes()).hexdigest() == \
            '238e277ad1bc97b96fb4cae3d6070abc24bb01c1'
    ar = eng.arena(3)
    expanded = np.nonzero(ar['FC'][:1] >= 0)[0]
    assert len(expanded) == 1
    # walk the tree: every expanded, visited node satisfies the sum rule
    stack, checked = [0], 0
    while stack:
        slot = stack.pop()
        fc = int(ar['FC'][slot])
        if fc < 0:
            continue
        nv = int(ar['NV'][slot])
        kids = ar['N'][fc:fc + nv]
        assert (kids > 0).all() and int(ar['N'][slot]) == 1 + int(kids.sum())
        checked += 1
        stack.extend(range(fc, fc + nv))
    assert checked > 200
    eng.close()


def _sum_rule(eng, game):
    """N(node) = 1 + sum N(children) over every expanded node of the game's tree; returns (#checked, N(root))."""
    ar = eng.arena(game)
    stack, checked = [0], 0
    while stack:
        slot = stack.pop()
        fc, nv = int(ar['FC'][slot]), int(ar['NV'][slot])
        if int(ar['K'][slot]) == 0 or fc < 0:
            continue
        kids = ar['N'][fc:fc + nv]
        assert (kids > 0).all() and int(ar['N'][slot]) == 1 + int(kids.sum())
        checked += 1
        stack.extend(range(fc, fc + nv))
    return checked, int(ar['N'][0])


@pytest.mark.parametrize('config', ['C4_gomoku15_800sims_512games', 'C3_connect4_400sims_512games'])
def test_full_size_batches_of_the_baseline_configs(config):
    """BASELINE.json configs[3] (its 512-games-per-GPU share) and configs[2] at FULL size through the production path
    (BatchedSelfPlay.for_network: hand-written net, hipGraphs, Dirichlet noise, tree reuse, slots refilled): the arena
    sizing at the configured batch is tested, not only benchmarked.  Size-independent properties: N(root) = n_playout
    after the first move, N(root) = carried + n_playout afterwards, the sum rule on sampled trees, pi from exact
    visit counts, no overflow flag, no dropped subtree, arena use below capacity."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(0)
    if config.startswith('C4'):
        net, kw, sims, moves = PolicyValueNet(15).to('cuda:0'), dict(board=15, n_in_row=5), 800, 6
    else:
        net = PolicyValueNet(6, 7, 7).to('cuda:0')
        kw, sims, moves = dict(board=(6, 7), n_in_row=4, game='connect4', net_shape=(6, 7, 7)), 400, 42
    G = 512
    sp = BatchedSelfPlay.for_network(net, n_games=G, n_playout=sims, lanes=1, seed=5, **kw)
    eng = sp.eng
    sp._start(range(G), range(G))
    sp._set_active()
    carried = np.zeros(G, dtype=np.int64)
    finished, worst_slots, worst_blocks = [], 0, 0
    for ply in range(moves):
        running = sp.slot_game >= 0
        if not running.any():
            break
        # what play_move() does, with the roots inspected between the search and the move
        sp._simulate()
        rn, _ = eng.root_stats()
        visits = eng.root_visits()
        assert (rn[running] == carried[running] + sims).all(), ply
        assert (visits.sum(axis=1)[running] == rn[running] - 1).all()  # N(root) = 1 + sum over its children
        for g in np.nonzero(running)[0][:3]:
            checked, n_root = _sum_rule(eng, int(g))
            assert n_root == rn[g] and checked >= 1
        # ... then the move itself, through the production code
        sp._simulate = lambda: None
        done = sp.play_move()
        del sp._simulate
        st = eng.check()
        assert st.reuse_dropped == 0
        worst_slots, worst_blocks = max(worst_slots, st.max_slots_used), max(worst_blocks, st.max_blocks_used)
        rn2, _ = eng.root_stats()
        carried = rn2.astype(np.int64)
        assert (carried[running] <= rn[running] - 1).all()  # the kept child's visits
        finished.extend(done)
        if done:
            sp.retire_finished()
            carried[sp.slot_game < 0] = 0
    st = eng.check()
    assert worst_slots < st.arena_slots and worst_blocks < st.arena_slots // 16
    for t in finished:
        assert abs(t.pis.sum(axis=1) - 1.0).max() < 1e-9 and t.winner in (-1, 0, 1)
    if config.startswith('C3'):
        assert len(finished) == G  # every Connect4 game ends within 42 plies
    for lane in sp.lanes:
        lane.eng.close()


@pytest.mark.parametrize('n_games,score_mode,forced_lanes', [(512, 'uct_ref', 0), (512, 'uct_ref', 4), (1536, 'uct_ref', 0), (1536, 'uct_ref', 2), (512, 'puct', 0), (256, 'uct_ref', 0)])
def test_full_size_shipped_layout_equals_one_plain_lane(n_games, score_mode, forced_lanes):
    """The layouts bench.py times at FULL size (15x15, 800 simulations per move) -- 512 games (BASELINE.json configs[3]'s share of a GPU)
    and 256: ONE lane of the resident search with the receptive-field trunk, two games per CU, one launch per search (k_delta_res);
    1536 games: the same single lane, its launch of 1536 workgroups running in three rounds of two per CU -- and (forced: the layout
    before the rounds) two lanes of the two-launch step with that trunk (k_trunk_delta), hipGraphs of 16 steps;
    512 games on FOUR such lanes (forced: the shipped layout of round 5); the opt-in PUCT rule: four lanes of the three-launch step
    with the full-board trunk -- the host side of a lane's move pipelined under the other lanes' simulations -- against ONE lane
    launched kernel by kernel with the FULL-BOARD trunk on every leaf (k_trunk_rows) and every move finished on the host before
    the next search: layout and receptive-field evaluation change no bit, so every game's moves and pi are the same; no subtree
    dropped, no flag, pi from exact counts."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(0)
    net = PolicyValueNet(15).to('cuda:0')
    plies, sims = 4, 800

    def play(shipped):
        # (256 games: the shipped layout is the RESIDENT search on two lanes -- one launch per search, a workgroup per game --; the
        # plain lane then runs the two-launch step kernel by kernel)
        kw = (dict(lanes=forced_lanes, resident_search=False) if forced_lanes else {}) if shipped else dict(lanes=1, use_graph=False, resident_search=False, delta_trunk=False)
        sp = BatchedSelfPlay.for_network(net, 15, 5, n_games=n_games, n_playout=sims, seed=5, score_mode=score_mode, **kw)
        if shipped:
            import rlzero_amd
            assert rlzero_amd.HW_QUEUES >= 8   # (claimed on import, before this process touched the GPU)
            one_resident_lane = score_mode == 'uct_ref' and not forced_lanes
            assert len(sp.lanes) == (forced_lanes or (1 if one_resident_lane else 4 if n_games == 512 else 2)) and sp.trunk_workgroups == 0 and sp.use_graph
            resident = [lane.evaluator.resident_ok(lane.eng) for lane in sp.lanes]
            assert resident == [one_resident_lane] * len(sp.lanes)
            assert all(lane.evaluator.delta_ok(lane.eng) for lane in sp.lanes)
            if not one_resident_lane:
                assert all(lane.evaluator.hip.heads_algo == 'parts' for lane in sp.lanes)
        else:
            assert not sp.lanes[0].evaluator.resident_ok(sp.lanes[0].eng) and not sp.lanes[0].evaluator.delta_ok(sp.lanes[0].eng)
        sp._start(range(n_games), range(n_games))
        sp._set_active()
        for _ in range(plies):
            done = sp.play_move_pipelined() if shipped else sp.play_move()
            assert not done   # no 15x15 game ends within 4 plies
        torch.cuda.synchronize()
        for st in sp.check():
            assert st.reuse_dropped == 0 and st.max_slots_used < st.arena_slots
        assert sp.moves_done == plies * n_games and sp.sims_done >= plies * n_games * sims
        moves = np.array([sp.slot_moves[s] for s in range(n_games)], dtype=np.int64)
        pis = np.array([sp.slot_pis[s] for s in range(n_games)], dtype=np.float64)
        for lane in sp.lanes:
            lane.evaluator.hip.check_flags()
            lane.eng.close()
        return moves, pis

    moves, pis = play(True)
    moves1, pis1 = play(False)
    assert moves.shape == (n_games, plies) and np.array_equal(moves, moves1)
    assert np.array_equal(pis.view(np.uint64), pis1.view(np.uint64))
    assert np.abs(pis.sum(axis=2) - 1.0).max() < 1e-9
    for ply in range(1, plies):   # a move never lands on an occupied cell, and pi gives earlier moves no mass
        for earlier in range(ply):
            assert (moves[:, ply] != moves[:, earlier]).all()
            assert (pis[np.arange(n_games), ply, moves[:, earlier]] == 0.0).all()


def test_whole_games_of_a_batch_in_rounds_equal_one_plain_lane():
    """A batch beyond two games per CU, whole games: `run_device(range(900))` with 800 slots at 13x13 / 200 simulations per move is ONE
    lane whose every search is one launch of 800 workgroups of k_delta_res -- a full round of 2 x CUs and a partial one, fewer as
    games end and the queue runs dry --, finished slots refilled on the device.  A sample of first-generation and refilled games
    against ONE plain lane launched kernel by kernel with the full-board trunk: moves, pi bits, winners."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(1)
    net = PolicyValueNet(13).to('cuda:0')
    n_slots, n_ids, sims = 800, 900, 200
    sample = sorted(set(list(range(0, 800, 23)) + list(range(800, 900, 9)) + [799, 899]))

    def play(shipped):
        kw = {} if shipped else dict(lanes=1, use_graph=False, resident_search=False, delta_trunk=False)
        sp = BatchedSelfPlay.for_network(net, 13, 5, n_games=n_slots if shipped else len(sample), n_playout=sims, seed=9, **kw)
        if shipped:
            n_cus = torch.cuda.get_device_properties(0).multi_processor_count
            assert len(sp.lanes) == 1 and sp.use_graph and (n_slots >= 3 * n_cus or n_cus != 256)
            assert sp.lanes[0].evaluator.resident_ok(sp.lanes[0].eng) and sp.lanes[0].evaluator.resident_delta_ok(sp.lanes[0].eng)
            out = sp.run_device(range(n_ids))
            assert sp.stalls_resolved == 0 and sp.sims_done == sims * sum(len(t.moves) for t in out)
        else:
            out = sp.run(sample)
        for st in sp.check():
            assert st.reuse_dropped == 0 and st.max_slots_used < st.arena_slots
        for lane in sp.lanes:
            lane.evaluator.hip.check_flags()
            lane.eng.close()
        return {t.game_id: t for t in out}

    shipped = play(True)
    assert sorted(shipped) == list(range(n_ids))
    plain = play(False)
    assert sorted(plain) == sorted(sample)
    for g in sample:
        a, b = shipped[g], plain[g]
        assert a.moves == b.moves and a.winner == b.winner, 'game %d depends on the layout' % g
        assert np.array_equal(a.pis.view(np.uint64), b.pis.view(np.uint64))
    for t in shipped.values():
        assert 9 <= len(t.moves) <= 169 and len(set(t.moves)) == len(t.moves) and t.winner in (-1, 0, 1)
        assert abs(t.pis.sum(axis=1) - 1.0).max() < 1e-9


def test_whole_games_on_the_shipped_layout_equal_one_plain_lane():
    """WHOLE games at full size on the layout bench.py times: `BatchedSelfPlay.run(range(640), pipelined=True)` and
    `run_device(range(640))` (the move step on the device: what bench.py runs by default) -- one lane of 512 games on the resident
    search with the receptive-field trunk (two games per CU, a whole move one hipGraph), finished slots refilled with the ids
    512 .. 639 -- at 15x15 / 800 simulations per move, every game played to its END (the reference's loop: game.py:96-134), against ONE
    plain lane launched kernel by kernel with the full-board trunk on a sample of 64 of the same ids (first-generation games and refilled ones): moves, pi bits and winners
    equal.  update_with_move keeps its subtree at every ply of every game (alphazero_mcts.py:96-103: the reference's tree is
    unbounded; here reuse_dropped counts a subtree over the carry limit and must stay 0) and no arena fills up."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(0)
    net = PolicyValueNet(15).to('cuda:0')
    n_ids, sims = 640, 800
    sample = list(range(0, 512, 11)) + list(range(512, 640, 8))       # 47 first-generation games + 16 refilled ones
    sample = sample + [511]
    assert len(sample) == 64

    def play(shipped, device_moves=False):
        kw = {} if shipped else dict(lanes=1, use_graph=False, resident_search=False, delta_trunk=False)
        sp = BatchedSelfPlay.for_network(net, 15, 5, n_games=512 if shipped else len(sample), n_playout=sims, seed=5, **kw)
        if shipped:
            assert len(sp.lanes) == 1 and sp.trunk_workgroups == 0 and sp.use_graph
            assert sp.lanes[0].evaluator.resident_ok(sp.lanes[0].eng) and sp.lanes[0].evaluator.resident_delta_ok(sp.lanes[0].eng)
        if device_moves:   # the move step on the device, as bench.py times it: the host reads the games from the log
            out = sp.run_device(range(n_ids))
            assert sp.stalls_resolved == 0 and sp.sims_done == sims * sum(len(t.moves) for t in out)
        else:
            out = sp.run(range(n_ids) if shipped else sample, pipelined=shipped)
        stats = sp.check()
        for st in stats:
            assert st.reuse_dropped == 0, 'a kept subtree exceeded the carry limit: not the reference\'s update_with_move'
            assert st.max_slots_used < st.arena_slots
        for lane in sp.lanes:
            lane.evaluator.hip.check_flags()
            lane.eng.close()
        return {t.game_id: t for t in out}, max(st.max_slots_used for st in stats), stats[0].arena_slots

    shipped, used, slots = play(True)
    assert sorted(shipped) == list(range(n_ids))
    on_device, used_d, _ = play(True, device_moves=True)
    assert sorted(on_device) == list(range(n_ids)) and used <= used_d < slots   # (device moves report the arenas' high-water mark)
    for g in range(n_ids):   # EVERY game of the device-driven run is the host-driven run's game
        a, b = shipped[g], on_device[g]
        assert a.moves == b.moves and a.winner == b.winner, 'game %d depends on where its moves are drawn' % g
        assert np.array_equal(a.pis.view(np.uint64), b.pis.view(np.uint64))
    plain, _, _ = play(False)
    assert sorted(plain) == sorted(sample)
    lengths = []
    for g in sample:
        a, b = shipped[g], plain[g]
        assert a.moves == b.moves and a.winner == b.winner, 'game %d depends on the layout' % g
        assert np.array_equal(a.pis.view(np.uint64), b.pis.view(np.uint64))
        lengths.append(len(a.moves))
    for t in shipped.values():                                          # every game: a legal, finished game of Gomoku
        assert 9 <= len(t.moves) <= 225 and len(set(t.moves)) == len(t.moves) and t.winner in (-1, 0, 1)
        assert t.winner != -1 or len(t.moves) == 225                    # a tie is a full board
        assert abs(t.pis.sum(axis=1) - 1.0).max() < 1e-9
    assert max(len(t.moves) for t in shipped.values()) > 60, 'the carry limit is only exercised by long games'
    print('whole games: mean %.1f plies, longest %d; fullest arena %d of %d slots' % (
        np.mean([len(t.moves) for t in shipped.values()]), max(len(t.moves) for t in shipped.values()), used, slots))


# ------------------------------------------------------------------ player / game loop (G3)
class _Injected(object):

    def __init__(self, us):
        self.us = list(us)
        self.real = np.random.choice

    def __call__(self, acts, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return np.asarray(acts)[cdf.searchsorted(self.us.pop(0), side='right')]


def test_selfplay_games_through_reference_api(g3):
    """GameControl.start_self_play + AlphaZeroPlayer (this repo's, on the GPU) reproduce the
    reference's games: moves, pi, winner, z, states, tree reuse."""
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.mcts import AlphaZeroPlayer
    for game in g3['selfplay']:
        inj = _Injected([unhex(p['u']) for p in game['plies']])
        np.random.choice = inj
        try:
            env = GomokuEnv(game['B'], game['n'])
            player = AlphaZeroPlayer(EVALS[game['eval']], n_playout=game['n_playout'], c_puct=5,
                                     is_selfplay=True)
            roots = []
            real = player.mcts.simulate

            def spy(e, temperature=1e-3, _real=real, _roots=roots, _player=player):
                acts, probs = _real(e, temperature)
                eng = _player.mcts._engine
                rn, rw = eng.root_stats()
                _roots.append((int(rn[0]), float(rw[0]).hex(), eng.root_visits()[0], eng.root_wsum()[0]))
                return acts, probs

            player.mcts.simulate = spy
            winner, data = GameControl(env).start_self_play(player, temperature=game['T'])
            data = list(data)
        finally:
            np.random.choice = inj.real
        assert winner == game['winner'] and list(env.states.keys()) == game['moves']
        assert len(data) == len(game['plies']) and not inj.us
        for (state, pi, z), root, ply in zip(data, roots, game['plies']):
            assert root[0] == ply['root_N'] and root[1] == ply['root_W']
            assert [int(root[2][a]) for a in ply['acts']] == ply['N']
            assert [float(root[3][a]).hex() for a in ply['acts']] == ply['W']
            assert bits_of_planes(state) == ply['obs'] and float(z) == ply['z']
            assert np.max(np.abs(pi - np.array([unhex(p) for p in ply['pi_full']]))) <= 1e-12
        view = player.mcts._root
        assert [view.explore_count, len(view._children)] == game['root_after_reset']
        player.mcts._engine.close()


def test_two_player_games_through_reference_api(g3):
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.mcts import AlphaZeroPlayer
    for duel in g3['duels']:
        inj = _Injected([unhex(u) for u in duel['u']])
        np.random.choice = inj
        try:
            env = GomokuEnv(duel['B'], duel['n'])
            p1 = AlphaZeroPlayer(ev.vlin, n_playout=duel['n_playout'][0], c_puct=5)
            p2 = AlphaZeroPlayer(ev.v0, n_playout=duel['n_playout'][1], c_puct=5)
            winner = GameControl(env).start_play(p1, p2, start_player=0, is_shown=0)
        finally:
            np.random.choice = inj.real
        assert winner == duel['winner'] and list(env.states.keys()) == duel['moves'] and not inj.us
        p1.mcts._engine.close()
        p2.mcts._engine.close()


def test_tree_view_matches_oracle_tree():
    from rlzero_amd.games import GomokuEnv
    from rlzero_amd.mcts.alphazero_mcts import AlphaZeroMCTS
    env = GomokuEnv(6, 4)
    env.reset()
    for m in (14, 15, 20):
        env.step(m)
    mcts = AlphaZeroMCTS(ev.vlin, n_playout=150, c_puct=5)
    acts, probs = mcts.simulate(env, 1.0)
    s = RefSearch(ev.vlin, 150, 5)
    s.simulate(RefGomoku.from_moves(6, 4, [14, 15, 20]), 1.0)
    root = mcts._root
    assert root.is_root() and not root.is_leaf() and root.explore_count == s.root.n
    assert list(root._children) == list(s.root.acts) == list(acts)
    for a, kid in zip(s.root.acts, s.root.kids):
        view = root._children[a]
        assert (view.explore_count, float(view.total_reward)) == (kid.n, float(kid.w))
        assert abs(view.prior - kid.p) < 1e-6
        assert len(view._children) == len(kid.kids)
    assert env.states == {14: 0, 15: 1, 20: 0}  # the caller's env is untouched
    mcts._engine.close()


# ------------------------------------------------------------------ batched self-play
def test_batched_selfplay_vs_oracle():
    """BatchedSelfPlay (lock-step games, device evaluator, counter-based uniforms) gives, game
    by game, the oracle's self-play game for the same uniforms; slots are refilled."""
    from rlzero_amd.engine import SyntheticEvaluator
    from rlzero_amd.selfplay import BatchedSelfPlay, move_uniform
    eng = _engine(6, 4, n_games=8, n_playout=60)
    sp = BatchedSelfPlay(eng, SyntheticEvaluator('vlin'), temperature=1.0, seed=5)
    trajs = sp.run(range(20))
    assert [t.game_id for t in trajs] == list(range(20))
    for t in trajs[::3]:
        us = move_uniform(5, np.full(64, t.game_id), np.arange(64))
        player = RefPlayer(ev.vlin, 60, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
        winner, data, moves = self_play_game(RefGomoku(6, 4), player, temperature=1.0)
        assert (winner, moves) == (t.winner, t.moves)
        w2, data2 = t.as_reference_tuple()
        for (s1, p1, z1), (s2, p2, z2) in zip(data, data2):
            assert (s1 == s2).all() and np.max(np.abs(p1 - p2)) <= 1e-12 and z1 == z2
    eng.close()


def test_two_capped_lanes_with_graphs_equal_one_eager_lane():
    """Lanes, hipGraph replay and the persistent-workgroup cap of the trunk are scheduling only: games are
    keyed by id, so 2 lanes x 6 games (graphs, trunk capped at 8 workgroups) give the very trajectories of 1
    lane x 12 games run kernel by kernel -- also after a weight update, which the captured graphs survive."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(3)
    net = PolicyValueNet(6).to('cuda:0').eval()
    kw = dict(board=6, n_in_row=4, n_games=12, n_playout=40, c_puct=5.0, device='cuda:0', temperature=1.0, seed=11)
    one = BatchedSelfPlay.for_network(net, lanes=1, use_graph=False, **kw)
    two = BatchedSelfPlay.for_network(net, lanes=2, trunk_workgroups=8, use_graph=True, sims_per_graph=8, **kw)
    assert len(two.lanes) == 2 and two.trunk_workgroups == 8

    def same(a, b):
        assert [t.game_id for t in a] == [t.game_id for t in b]
        for x, y in zip(a, b):
            assert (x.winner, x.moves) == (y.winner, y.moves)
            assert np.array_equal(np.asarray(x.pis), np.asarray(y.pis))

    same(one.run(range(30)), two.run(range(30)))
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.25)
    one.refresh_weights()
    two.refresh_weights()
    a, b = one.run(range(30, 54)), two.run(range(30, 54))
    same(a, b)
    # pipelined moves (a lane's host step under the other lanes' simulations, finished slots refilled lane by lane, the
    # default co-resident layout: un-capped trunks + 'parts' GEMM) and three lanes: still the same trajectories
    three = BatchedSelfPlay.for_network(net, lanes=3, use_graph=True, sims_per_graph=8, **kw)
    assert three.lanes[0].evaluator.hip.max_boards >= 4 and three.trunk_workgroups == 0
    same(one.run(range(54, 90)), three.run(range(54, 90), pipelined=True))
    same(two.run(range(90, 110), pipelined=True), three.run(range(90, 110)))
    # four lanes (the layout of the BASELINE batch, each lane on a hardware queue of its own) with refills lane by lane
    four = BatchedSelfPlay.for_network(net, lanes=4, use_graph=True, sims_per_graph=8, **kw)
    assert len(four.lanes) == 4 and [lane.eng.n_games for lane in four.lanes] == [3, 3, 3, 3]
    same(one.run(range(400, 440)), four.run(range(400, 440), pipelined=True))
    # a pipelined run cut short leaves lanes with the next move's simulations enqueued: a later run starts clean, and
    # play_move() after play_move_pipelined() takes a primed lane's search as it is (no second search on top)
    three.run(range(200, 236), max_moves=3, pipelined=True)
    assert any(getattr(lane, 'primed', False) for lane in three.lanes)
    same(one.run(range(110, 140)), three.run(range(110, 140)))
    # ... also when the later run has fewer games than slots: the games the cut-short run left behind are abandoned (not played
    # on, not searched a second time on top of their finished search, not returned)
    three.run(range(236, 272), max_moves=2, pipelined=True)
    assert (three.slot_game >= 0).sum() == 12
    few = three.run(range(140, 145), pipelined=True)
    same(one.run(range(140, 145)), few)
    assert [t.game_id for t in few] == list(range(140, 145)) and (three.slot_game < 0).all()
    # the PUCT rule READS the priors, Dirichlet noise included: a game's noise stream is keyed by (seed, game id)
    # (rz_set_noise_keys), so slots, lanes and refills still do not matter
    pk = dict(kw, score_mode='puct')
    p_one = BatchedSelfPlay.for_network(net, lanes=1, use_graph=False, **pk)
    p_three = BatchedSelfPlay.for_network(net, lanes=3, use_graph=True, sims_per_graph=8, **pk)
    a, b = p_one.run(range(500, 530)), p_three.run(range(500, 530), pipelined=True)
    same(a, b)
    same(a[:6], p_one.run(range(500, 506)))   # ... nor the games played before in the same slots
    assert [t.moves for t in a] != [t.moves for t in one.run(range(500, 530))]   # (the rule does change the games)
    mixed = BatchedSelfPlay.for_network(net, lanes=2, use_graph=True, sims_per_graph=8, **kw)
    plain = BatchedSelfPlay.for_network(net, lanes=1, use_graph=False, **kw)
    for sp_ in (mixed, plain):
        sp_._start(range(12), range(300, 312))
        sp_._set_active()
    for ply in range(4):
        assert not (mixed.play_move_pipelined() if ply % 2 == 0 else mixed.play_move()) and not plain.play_move()
    assert mixed.slot_moves == plain.slot_moves
    rn_m = np.concatenate([lane.eng.root_stats()[0] for lane in mixed.lanes])
    assert np.array_equal(rn_m, plain.eng.root_stats()[0])   # no tree carries a search too many
    for lane in one.lanes + two.lanes + three.lanes + mixed.lanes + plain.lanes:
        lane.eng.close()


# ------------------------------------------------------------------ real net
def test_replay_recorded_leaf_values(g2net):
    """Feeding the reference's recorded per-simulation leaf values (real net on CPU) through
    the device tree rebuilds the reference's tree bit-for-bit."""
    from rlzero_amd.engine import HostEvaluator
    for rec in g2net['cases']:
        B, n = rec['B'], rec['n']
        values = iter([unhex(v) for _, v in rec['leaves']])
        boards = iter([m for m, _ in rec['leaves']])

        def replay(env, _values=values, _boards=boards):
            assert sorted(env.states.keys()) == sorted(next(_boards))
            legal = env.leagel_actions()
            return [(a, 1.0 / max(len(legal), 1)) for a in legal], next(_values)

        env = RefGomoku.from_moves(B, n, rec['pre'])
        eng = _engine(B, n, n_games=1, n_playout=rec['n_playout'])
        _set_roots(eng, [env], reset_trees=True)
        host = HostEvaluator(replay, lambda s0, s1, tm, last, B=B, n=n: _make_env(B, n, s0, s1, tm, last))
        rec = dict(rec, n_nodes=len(rec['tree']))
        _check_case(eng, host, rec, env)
        eng.close()


def _make_env(B, n, s0, s1, tm, last):
    from rlzero_amd.games import GomokuEnv
    return GomokuEnv.from_bitboards(B, n, s0, s1, tm, last)


def test_cpu_agent_through_host_path_vs_oracle():
    """A CPU AlphaZeroAgent.policy_value_fn is an arbitrary callable for the engine: the
    search must equal the oracle's search driven by the very same callable."""
    import torch
    torch.set_num_threads(1)
    from rlzero_amd.games import GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts.alphazero_mcts import AlphaZeroMCTS
    agent = AlphaZeroAgent(6, device='cpu')
    agent.policy_value_net.load_state_dict({k: torch.from_numpy(v) for k, v in ev.numpy_weights(6, 101).items()})
    env = GomokuEnv(6, 4)
    env.reset()
    for m in (14, 15, 20):
        env.step(m)
    with torch.no_grad():
        mcts = AlphaZeroMCTS(agent.policy_value_fn, n_playout=120, c_puct=5)
        acts, probs = mcts.simulate(env, 1.0)
        s = RefSearch(agent.policy_value_fn, 120, 5)
        acts2, probs2 = s.simulate(RefGomoku.from_moves(6, 4, [14, 15, 20]), 1.0)
    assert list(acts) == list(acts2) and np.max(np.abs(probs - probs2)) <= 1e-12
    assert _hex_tree(mcts._engine.tree_dump(0)) == _hex_tree(tree_dump(s.root))
    mcts._engine.close()


def test_net_on_gpu_vs_golden(g4):
    """policy/value of PolicyValueNet on the MI355X vs the reference's CPU outputs: 1e-4."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    for B in (3, 6, 9, 15):
        net = PolicyValueNet(B)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in ev.numpy_weights(B, int(g4['B%d_seed' % B])).items()})
        net = net.to('cuda:0')
        obs = torch.from_numpy(g4['B%d_obs' % B].astype(np.float32)).to('cuda:0')
        with torch.no_grad():
            logp, v = net(obs)
        assert np.max(np.abs(logp.cpu().numpy() - g4['B%d_logp' % B])) <= 1e-4
        assert np.max(np.abs(v.cpu().numpy() - g4['B%d_value' % B])) <= 1e-4


def test_gpu_agent_fast_path_selfplay_game():
    """AlphaZeroAgent on the GPU -> the batched device evaluator is used; a whole self-play
    game runs through the reference API and satisfies the search invariants."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts import AlphaZeroPlayer
    torch.manual_seed(0)
    np.random.seed(0)
    agent = AlphaZeroAgent(6, device='cuda:0')
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=100, c_puct=5, is_selfplay=True)
    env = GomokuEnv(6, 4)
    winner, data = GameControl(env).start_self_play(player, temperature=1.0)
    data = list(data)
    assert isinstance(player.mcts._evaluator, HipNetEvaluator)
    assert winner in (-1, 0, 1) and len(data) == len(env.states) >= 7
    for state, pi, z in data:
        assert abs(pi.sum() - 1.0) < 1e-9 and state.shape == (4, 6, 6)
    # the net's value reaches the tree: leaf value of the first simulation = net(root obs)
    env.reset()
    mcts = player.mcts
    mcts.n_playout = 1
    mcts.update_with_move(-1)
    mcts.simulate(env, 1.0)
    rn, rw = mcts._engine.root_stats()
    with torch.no_grad():
        _, v = agent.policy_value_net(torch.from_numpy(env.current_state()[None]).float().to('cuda:0'))
    assert int(rn[0]) == 1 and abs(rw[0] + float(v.item())) <= 1e-5
    # a training step changes the weights: the next search must see them
    states = [s for s, _, _ in data][:8]
    pis = [p for _, p, _ in data][:8]
    agent.learn(states, pis, [1.0] * len(states))
    mcts.update_with_move(-1)
    mcts.simulate(env, 1.0)
    _, rw2 = mcts._engine.root_stats()
    with torch.no_grad():
        _, v2 = agent.policy_value_net(torch.from_numpy(env.current_state()[None]).float().to('cuda:0'))
    assert abs(rw2[0] + float(v2.item())) <= 1e-5 and abs(float(v2.item()) - float(v.item())) > 1e-6
    player.mcts._engine.close()


# ------------------------------------------------------------------ hand-written net forward
def test_hip_net_vs_golden_and_torch(g4):
    """csrc/rz_net.hip (fused fp32 MFMA forward) vs the reference's CPU outputs (1e-4, the
    tolerance of BASELINE.json) and vs the torch module on the same GPU, all board sizes."""
    import torch
    from rlzero_amd.engine import HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    for B, algo in [(b, a) for b in (3, 6, 9, 15) for a in ('winograd_f4', 'direct', 'split_f16')]:
        weights = ev.numpy_weights(B, int(g4['B%d_seed' % B]))
        hip = HipNet(B, 'cuda:0', max_boards=16).load_state_dict(weights).set_algo(algo)
        obs = torch.from_numpy(g4['B%d_obs' % B].astype(np.float32)).to('cuda:0')
        logp, value = hip.forward(obs)
        assert np.max(np.abs(logp.cpu().numpy() - g4['B%d_logp' % B])) <= 1e-4
        assert np.max(np.abs(value.cpu().numpy()[:, None] - g4['B%d_value' % B])) <= 1e-4
        assert abs(float(torch.exp(logp).sum(1).mean()) - 1.0) < 1e-5
        # trunk features against torch fp64 on random (non 0/1) inputs
        rs = np.random.RandomState(B)
        x = torch.from_numpy(rs.standard_normal((37, 4, B, B)).astype(np.float32)).to('cuda:0')
        lp64, v64 = ev.net_forward(weights, x.cpu().numpy(), dtype=torch.float64)
        lp, v = hip.forward(x)
        assert np.max(np.abs(lp.cpu().numpy() - lp64.numpy())) <= 1e-4
        assert np.max(np.abs(v.cpu().numpy() - v64.numpy()[:, 0])) <= 1e-4
        if algo != 'direct':
            # persistent workgroups capped at 5: each loops over up to 8 of the 37 boards (ragged tail), with
            # the next board's planes prefetched -- must be bit-identical to one board per workgroup
            lp5, v5 = hip.set_max_workgroups(5).forward(x)
            assert torch.equal(lp5, lp) and torch.equal(v5, v)
            hip.set_max_workgroups(0)
        hip.close()
    # A = I style layout check: an asymmetric single-tap kernel must shift, not transpose
    B = 6
    w = {k: np.zeros_like(v) for k, v in ev.numpy_weights(B, 1).items()}
    w['conv1.weight'][5, 2, 0, 2] = 1.0   # out ch 5 <- in ch 2 shifted by (dy=-1, dx=+1)
    w['conv2.weight'][7, 5, 1, 1] = 1.0
    w['conv3.weight'][9, 7, 2, 0] = 1.0   # (dy=+1, dx=-1)
    w['act_conv1.weight'][1, 9, 0, 0] = 1.0
    hip = HipNet(B, 'cuda:0', max_boards=4).load_state_dict(w)
    x = torch.rand((2, 4, B, B), device='cuda:0')
    feat = hip.trunk(x).cpu().numpy().reshape(2, 6, B, B)
    feat_direct = hip.set_algo('direct').trunk(x).cpu().numpy().reshape(2, 6, B, B)
    assert np.max(np.abs(feat - feat_direct)) <= 1e-5
    net = PolicyValueNet(B)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    with torch.no_grad():
        t = x.cpu()
        for conv in (net.conv1, net.conv2, net.conv3):
            t = torch.relu(conv(t))
        want = torch.relu(net.act_conv1(t)).numpy()
    assert np.max(np.abs(feat[:, :4] - want)) <= 1e-6 and np.abs(want).max() > 0.1
    hip.close()


def test_split_trunk_every_board_size():
    """k_trunk_split over all its shapes -- tiles of 2 x 16 (11 .. 16), one tile per wave (9, 10), the channel-split
    variants for boards of two tiles (6, 7, 8) and of one (3, 4, 5) -- against torch fp64 (1e-4, random non-0/1 inputs,
    37 boards on 5 persistent workgroups and on one each: same bits) and against the direct f32 kernel's features."""
    import torch
    from rlzero_amd.engine import HipNet
    for B in range(3, 17):
        weights = ev.numpy_weights(B, 1000 + B)
        hip = HipNet(B, 'cuda:0', max_boards=64).load_state_dict(weights)
        rs = np.random.RandomState(B)
        x = torch.from_numpy(rs.standard_normal((37, 4, B, B)).astype(np.float32)).to('cuda:0')
        lp64, v64 = ev.net_forward(weights, x.cpu().numpy(), dtype=torch.float64)
        lp, v = hip.forward(x)
        assert np.max(np.abs(lp.cpu().numpy() - lp64.numpy())) <= 1e-4, B
        assert np.max(np.abs(v.cpu().numpy() - v64.numpy()[:, 0])) <= 1e-4, B
        lp5, v5 = hip.set_max_workgroups(5).forward(x)
        assert torch.equal(lp5, lp) and torch.equal(v5, v), B
        hip.set_max_workgroups(0)
        planes = (torch.rand((9, 4, B, B), device='cuda:0') < 0.4).float()   # 0 / 1 planes like the tree's leaves
        feat = hip.trunk(planes).cpu().numpy()
        feat_direct = hip.set_algo('direct').trunk(planes).cpu().numpy()
        assert np.max(np.abs(feat - feat_direct)) <= 2e-5 * max(1.0, float(np.abs(feat_direct).max())), B
        hip.check_flags()
        hip.close()


def test_split_trunk_rectangular_boards():
    """Rectangular boards through every tile shape of k_trunk_split -- (7, 9) / (8, 10) / (9, 10): three 3-row tiles = the
    3 + 1 channel split; (6, 7) / (12, 5) / (4, 16): two tiles = two channel halves; (2, 13) / (5, 6): one tile = four
    channel quarters; (10, 10) / (8, 3): four tiles; (13, 16) / (16, 11) / (5, 16) x ... : tiles of 2 x 16 -- against the
    torch module in fp64 (1e-4), ragged batches on capped workgroups giving the same bits."""
    import torch
    from rlzero_amd.engine import HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    shapes = [(7, 9), (8, 10), (9, 10), (6, 7), (12, 5), (4, 16), (2, 13), (5, 6), (10, 10), (8, 3), (13, 16), (16, 11), (11, 12)]
    for i, (rows, cols) in enumerate(shapes):
        torch.manual_seed(100 + i)
        n_actions = rows * cols
        net = PolicyValueNet(rows, cols, n_actions)
        hip = HipNet((rows, cols, n_actions), 'cuda:0', max_boards=64).load_state_dict(net.state_dict())
        x = torch.randn(35, 4, rows, cols)
        with torch.no_grad():
            lp64, v64 = net.double()(x.double())
        lp, v = hip.forward(x.to('cuda:0'))
        assert np.max(np.abs(lp.cpu().numpy() - lp64.numpy())) <= 1e-4, (rows, cols)
        assert np.max(np.abs(v.cpu().numpy() - v64.numpy()[:, 0])) <= 1e-4, (rows, cols)
        lp3, v3 = hip.set_max_workgroups(3).forward(x.to('cuda:0'))
        assert torch.equal(lp3, lp) and torch.equal(v3, v), (rows, cols)
        hip.check_flags()
        hip.close()


def test_row_tile_trunk_on_its_board_shapes():
    """k_trunk_rows (boards of 11 .. 16 rows and columns: v_mfma_f32_16x16x32_f16, one N-tile per board row, the waves
    split the output channels) against the torch module in fp64 (1e-4 on log-probabilities and values, random non-0/1
    inputs) and against k_trunk_split on the same boards ('split_f16_tiles': the same hi + lo arithmetic in another
    summation order -- f32 accumulation rounding apart); ragged batches on capped workgroups give the same bits.  Every
    width at 15 rows, every row count at some width."""
    import torch
    from rlzero_amd.engine import HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    shapes = [(15, cols) for cols in range(11, 17)] + [(11, 11), (12, 16), (13, 13), (14, 12), (16, 11), (16, 16)]
    for i, (rows, cols) in enumerate(shapes):
        torch.manual_seed(300 + i)
        n_actions = rows * cols
        net = PolicyValueNet(rows, cols, n_actions)
        hip = HipNet((rows, cols, n_actions), 'cuda:0', max_boards=64).load_state_dict(net.state_dict())
        x = torch.randn(37, 4, rows, cols)
        with torch.no_grad():
            lp64, v64 = net.double()(x.double())
        xd = x.to('cuda:0')
        lp, v = hip.forward(xd)
        assert np.max(np.abs(lp.cpu().numpy() - lp64.numpy())) <= 1e-4, (rows, cols)
        assert np.max(np.abs(v.cpu().numpy() - v64.numpy()[:, 0])) <= 1e-4, (rows, cols)
        lp3, v3 = hip.set_max_workgroups(3).forward(xd)
        assert torch.equal(lp3, lp) and torch.equal(v3, v), (rows, cols)
        hip.set_max_workgroups(0)
        planes = (torch.rand((9, 4, rows, cols), device='cuda:0') < 0.4).float()   # 0 / 1 planes like the tree's leaves
        feat = hip.trunk(planes).cpu().numpy()
        feat_tiles = hip.set_algo('split_f16_tiles').trunk(planes).cpu().numpy()
        feat_direct = hip.set_algo('direct').trunk(planes).cpu().numpy()
        scale = max(1.0, float(np.abs(feat_direct).max()))
        assert np.abs(feat).max() > 0.02 and np.max(np.abs(feat - feat_tiles)) <= 2e-6 * scale, (rows, cols)
        assert np.max(np.abs(feat - feat_direct)) <= 2e-5 * scale, (rows, cols)
        hip.check_flags()
        hip.close()


def test_trunk_is_deterministic_under_load():
    """The default trunk accumulates with an inline-assembly MFMA whose register hazards are kept by hand
    (DESIGN.md section 7): a violated hazard would be timing dependent, so 60 launches of a full 512-board batch,
    alone and beside a second stream that keeps the small kernels busy, must give bit-identical features that
    also agree with the direct kernel."""
    import torch
    from rlzero_amd.engine import HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(11)
    net = PolicyValueNet(15)
    hip = HipNet(15, 'cuda:0', max_boards=512).load_state_dict(net.state_dict())
    x = (torch.rand((512, 4, 15, 15), device='cuda:0') > 0.5).float()
    ref = hip.set_algo('direct').trunk(x).clone()
    for algo, caps in (('split_f16', (0, 224)), ('winograd_f4', (0, 224))):
        hip.set_algo(algo)
        first = None
        side = torch.cuda.Stream()
        busy = torch.rand((2048, 2048), device='cuda:0')
        for it in range(60):
            hip.set_max_workgroups(caps[it % len(caps)])
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    busy = (busy @ busy).clamp_(-1, 1)
            out = hip.trunk(x)
            if first is None:
                first = out.clone()
                assert float((first - ref).abs().max()) < 5e-6
            else:
                assert torch.equal(out, first)
        torch.cuda.synchronize()
    hip.close()


def test_split_f16_trunk_is_as_accurate_as_the_f32_trunk_and_flags_its_range():
    """RZ_NET_SPLIT_F16 (f32 operands carried as hi + lo f16 pairs on the f16 matrix pipe, f32 accumulation):
    its error against an fp64 evaluation of the same weights is at the level of the exact-f32 direct kernel
    (and below the Winograd kernels'), on unit-scale and on 30x larger weights; an activation beyond the
    range of the scaled f16 pieces is reported, not silently wrong."""
    import torch
    from rlzero_amd.engine import HipError, HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    for B, gain in ((15, 1.0), (15, 30.0), (9, 0.03), (16, 1.0)):
        torch.manual_seed(5)
        net = PolicyValueNet(B)
        with torch.no_grad():
            net.conv2.weight.mul_(gain)
            net.conv3.weight.mul_(gain)
        weights = {k: v.detach().numpy() for k, v in net.state_dict().items()}
        hip = HipNet(B, 'cuda:0', max_boards=32).load_state_dict(weights)
        x = (torch.rand((24, 4, B, B), device='cuda:0') < 0.4).float()
        net64 = PolicyValueNet(B).double()
        net64.load_state_dict({k: torch.from_numpy(v).double() for k, v in weights.items()})
        with torch.no_grad():
            h = x.cpu().double()
            for conv in (net64.conv1, net64.conv2, net64.conv3):
                h = torch.relu(conv(h))
            ref = torch.cat([torch.relu(net64.act_conv1(h)).flatten(1), torch.relu(net64.val_conv1(h)).flatten(1)], 1).numpy()
        err = {}
        for algo in ('direct', 'winograd_f4', 'split_f16'):
            out = hip.set_algo(algo).trunk(x).cpu().numpy().reshape(ref.shape)
            err[algo] = float(np.abs(out - ref).max()) / float(np.abs(ref).max())
        hip.check_flags()
        assert err['split_f16'] <= 1e-6, err
        assert err['split_f16'] <= 2.0 * err['direct'] + 1e-7, err
        assert err['split_f16'] <= err['winograd_f4'] + 1e-7, err
        hip.close()
    # weights of ANY scale stay in range on 0 / 1 planes: the activation scales follow the bounds rz_net_load derives.
    # conv1 x 1e5 (activations ~1e5: 16x that is far beyond f16), conv2 x 1e-5 brings the rest back to order 1
    torch.manual_seed(5)
    net = PolicyValueNet(6)
    with torch.no_grad():
        net.conv1.weight.mul_(1e5)
        net.conv1.bias.mul_(1e5)
        net.conv2.weight.mul_(1e-5)
    hip = HipNet(6, 'cuda:0', max_boards=8).load_state_dict(net.state_dict()).set_algo('split_f16')
    info = hip.range_info()
    assert info['split_ok'] and info['scales'][0] < 1.0 and info['bounds'][0] * info['scales'][0] < 60000.0
    assert info['scales'][1] == 16.0 and info['bounds'][0] > 1e4
    x = (torch.rand((4, 4, 6, 6), device='cuda:0') < 0.5).float()
    lp, v = hip.forward(x)
    hip.check_flags()
    with torch.no_grad():
        lp64, v64 = net.double()(x.cpu().double())
    assert float((lp.cpu().double() - lp64).abs().max()) <= 1e-4 and float((v.cpu().double() - v64[:, 0]).abs().max()) <= 1e-4
    lp_d, v_d = hip.set_algo('direct').forward(x)
    assert float((lp - lp_d).abs().max()) <= 2e-5  # the exact-f32 kernel is no closer to fp64 than the split one
    assert float((lp_d.cpu().double() - lp64).abs().max()) <= 1e-4
    # an input far outside [0, 1] is what can still overflow: reported, not silently wrong
    hip.set_algo('split_f16').trunk(1e6 * x)
    with pytest.raises(HipError):
        hip.check_flags()
    hip.check_flags()  # the flag is cleared by the report
    hip.close()
    # weights without finite bounds: the net runs on the exact-f32 direct kernel (no f16 piece is ever formed)
    net = PolicyValueNet(6)
    with torch.no_grad():
        net.conv2.weight[3, 2, 1, 1] = float('inf')
    hip = HipNet(6, 'cuda:0', max_boards=8).load_state_dict(net.state_dict())
    assert not hip.range_info()['split_ok']
    a = hip.trunk(x).clone()
    b = hip.set_algo('direct').trunk(x)
    assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    hip.check_flags()
    hip.close()


def test_scaled_up_weights_through_the_reference_api():
    """A net whose conv1 activations are far outside the f16 range (weights x 1e5, the next layer x 1e-5) played
    through AlphaZeroPlayer / GameControl on the default split-f16 trunk: no exception, and the priors / value
    that reach the tree agree with the torch module to the 1e-4 of the contract (alphazero_agent.py:31-46)."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts import AlphaZeroPlayer
    torch.manual_seed(2)
    np.random.seed(2)
    agent = AlphaZeroAgent(6, device='cuda:0')
    with torch.no_grad():
        agent.policy_value_net.conv1.weight.mul_(1e5)
        agent.policy_value_net.conv1.bias.mul_(1e5)
        agent.policy_value_net.conv2.weight.mul_(1e-5)
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=60, c_puct=5, is_selfplay=True)
    env = GomokuEnv(6, 4)
    winner, data = GameControl(env).start_self_play(player, temperature=1.0)
    assert isinstance(player.mcts._evaluator, HipNetEvaluator) and winner in (-1, 0, 1)
    assert player.mcts._evaluator.hip.range_info()['scales'][0] < 1.0
    # one expansion of a mid-game root: stored priors and backed-up value vs the torch module in fp64
    env.reset()
    for m in (14, 15, 20):
        env.step(m)
    mcts = player.mcts
    mcts.add_noise = False
    mcts._engine.close()
    mcts._engine = None
    mcts.n_playout = 1
    mcts.simulate(env, 1.0)
    pri = mcts._engine.root_priors()[0]
    _, rw = mcts._engine.root_stats()
    with torch.no_grad():
        lp64, v64 = agent.policy_value_net.double()(torch.from_numpy(env.current_state()[None]).double().to('cuda:0'))
    legal = env.leagel_actions()
    assert np.max(np.abs(np.log(pri[legal].astype(np.float64)) - lp64.cpu().numpy()[0][legal])) <= 1e-4
    assert abs(rw[0] + float(v64.item())) <= 1e-4
    mcts._engine.close()


def test_split_f16_heads_gemm_equals_the_f32_gemm():
    """k_heads_split (the FC GEMM on the f16 matrix pipe, hi + lo operand pairs, fed by the f16 feature pieces the
    split-f16 trunk writes) against the f32-input MFMA GEMM and an fp64 evaluation: ragged batches (partial 32- and
    64-board tiles), every board shape (1 .. 8 policy output tiles), both workgroup shapes bit-identical."""
    import torch
    from rlzero_amd.engine import HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    for shape, batches in ((15, (1, 33, 100, 672)), (16, (65, )), (9, (64, 70)), (3, (5, )), ((6, 7, 7), (47, ))):
        torch.manual_seed(3)
        net = PolicyValueNet(*shape) if isinstance(shape, tuple) else PolicyValueNet(shape)
        with torch.no_grad():  # away from the tiny default initialisation: logits of order 1
            net.act_fc1.weight.mul_(8.0)
            net.val_fc1.weight.mul_(8.0)
        hip = HipNet(shape, 'cuda:0', max_boards=max(batches)).load_state_dict(net.state_dict())
        net64 = (PolicyValueNet(*shape) if isinstance(shape, tuple) else PolicyValueNet(shape)).double()
        net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
        rows, cols = (shape[0], shape[1]) if isinstance(shape, tuple) else (shape, shape)
        for n in batches:
            x = (torch.rand((n, 4, rows, cols), device='cuda:0') < 0.4).float()
            out = {}
            for algo in ('f32', 'split32', 'split64', 'parts', 'auto', 'in_trunk'):
                lp, v = hip.set_heads_algo(algo).forward(x)
                out[algo] = (lp.clone(), v.clone())
            lp, v = hip.set_max_workgroups(8).forward(x)  # 'auto' beside a capped trunk: 64-board workgroups
            out['auto_capped'] = (lp.clone(), v.clone())
            hip.set_max_workgroups(0)
            hip.check_flags()
            with torch.no_grad():
                lp64, v64 = net64(x.cpu().double())
            for algo in ('f32', 'split32'):
                assert float((out[algo][0].cpu().double() - lp64).abs().max()) <= 2e-5, (shape, n, algo)
                assert float((out[algo][1].cpu().double() - v64[:, 0]).abs().max()) <= 2e-6, (shape, n, algo)
            assert float((out['f32'][0] - out['split32'][0]).abs().max()) <= 1e-5
            assert float((out['f32'][1] - out['split32'][1]).abs().max()) <= 2e-6
            # the workgroup shape / who adds the K quarters / the trunk's workgroups doing the layers on their own boards (boards of
            # up to 10 rows, 'in_trunk' and 'auto' up to one board per CU) is scheduling only: same bits
            for algo in ('split64', 'parts', 'auto', 'in_trunk', 'auto_capped'):
                assert torch.equal(out[algo][0], out['split32'][0]) and torch.equal(out[algo][1], out['split32'][1])
        # the split trunk wrote only the f16 pieces (GEMM 'auto'): a GEMM forced to f32 afterwards runs on them
        hip.set_heads_algo('auto').trunk_internal(x)
        lp_late = torch.empty_like(out['split32'][0])
        v_late = torch.empty_like(out['split32'][1])
        hip.set_heads_algo('f32').heads(x.shape[0], lp_late, v_late)
        assert torch.equal(lp_late, out['split32'][0]) and torch.equal(v_late, out['split32'][1])
        # after another trunk the f16 pieces are stale: the GEMM must take the f32 features
        lp_d, v_d = hip.set_algo('direct').set_heads_algo('split64').forward(x)
        lp_f, v_f = hip.set_heads_algo('f32').forward(x)
        assert torch.equal(lp_d, lp_f) and torch.equal(v_d, v_f)
        hip.close()


def test_search_with_hip_net_equals_search_with_its_values():
    """Tree built with the HIP net evaluator == oracle tree fed the very same fp32 values
    (the net's value reaches the tree unchanged; a 512-game batch runs clean)."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(0)
    net = PolicyValueNet(6).to('cuda:0')
    evaluator = HipNetEvaluator(net, 6, 'cuda:0', max_boards=8)
    evaluator.deferred_priors = False   # (values from the FC GEMM, like hip.forward below; the deferred route: tests/test_deferred.py)
    eng = _engine(6, 4, n_games=8, n_playout=80)
    eng.reset_games()
    eng.simulate(evaluator, 80)
    eng.check()
    rn, _ = eng.root_stats()
    assert (rn == 80).all()
    # game 0 from the empty board: recompute every leaf value with the same kernel, batch 1
    values = {}

    def pvf(env):
        obs = torch.from_numpy(env.current_state()[None]).float().to('cuda:0')
        logp, v = evaluator.hip.forward(obs)
        legal = env.leagel_actions()
        return [(a, 1.0 / len(legal)) for a in legal], float(v.item())

    s = RefSearch(pvf, 80, 5)
    s.simulate(RefGomoku(6, 4), 1.0)
    assert _hex_tree(eng.tree_dump(0)) == _hex_tree(tree_dump(s.root))
    eng.close()


class _UnfusedHipNet(object):
    """The same HipNet through the un-fused route: rz_net_forward (trunk + GEMM + k_heads_finish) writes log-probabilities
    and values, rz_expand_backup / rz_tree_step take exp() of them."""
    needs_obs = True

    def __init__(self, hip):
        self.hip = hip

    def __call__(self, eng):
        return self.hip.forward(eng.obs, eng.logp, eng.value)


def test_fused_route_priors_vs_golden_and_unfused(g4):
    """The production route (k_trunk_split -> k_heads_split -> k_tree_step_raw finishing log_softmax / tanh itself):
    the priors it stores and the value it backs up for G4's position (after move 0), against the reference's
    policy_value_fn output (alphazero_agent.py:41-45; 1e-4) and bit-for-bit against the un-fused rz_net_heads route;
    all trunk algorithms, all board sizes; and on a batch of random mid-game positions the two routes agree bit-for-bit."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    for B in (3, 6, 9, 15):
        n = 3 if B == 3 else (4 if B == 6 else 5)
        net = PolicyValueNet(B)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in ev.numpy_weights(B, int(g4['B%d_seed' % B])).items()})
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=16)
        evaluator.deferred_priors = False   # the route that writes priors inside the tree step (the deferred one: tests/test_deferred.py)
        acts = [int(a) for a in g4['B%d_pvf_acts' % B]]
        want_p, want_v = g4['B%d_pvf_probs' % B].astype(np.float64), float(g4['B%d_pvf_value' % B])
        rs = np.random.RandomState(B)
        envs = [RefGomoku.from_moves(B, n, [0])]
        while len(envs) < 16:  # random non-terminal mid-game positions
            e = RefGomoku(B, n)
            for m in rs.permutation(B * B)[:rs.randint(0, B * B - 1)]:
                e.step(int(m))
                if e.game_end_winner()[0]:
                    break
            if not e.game_end_winner()[0]:
                envs.append(e)
        for algo, heads in (('split_f16', 'auto'), ('split_f16', 'parts'), ('split_f16_tiles', 'auto'), ('winograd_f4', 'auto'), ('direct', 'auto')):
            evaluator.hip.set_algo(algo).set_heads_algo(heads)
            got = {}
            planes = HipNetEvaluator.__new__(HipNetEvaluator)  # the same HipNet, fed float planes instead of positions
            planes.module, planes.hip, planes.use_positions = evaluator.module, evaluator.hip, False
            assert planes.needs_obs and evaluator.needs_obs == (algo not in ('split_f16', 'split_f16_tiles'))
            for route, evl in (('fused', evaluator), ('fused_planes', planes), ('unfused', _UnfusedHipNet(evaluator.hip))):
                eng = _engine(B, n, n_games=len(envs), n_playout=4)
                _set_roots(eng, envs, reset_trees=True)
                eng.sim_chunk(evl, 1)  # the first simulation expands every root
                pri = eng.root_priors()
                rn, rw = eng.root_stats()
                eng.check()
                assert (rn == 1).all()
                got[route] = (pri.copy(), rw.copy())
                eng.close()
            pri, rw = got['fused']
            assert np.max(np.abs(pri[0][acts].astype(np.float64) - want_p)) <= 1e-4, (B, algo)
            assert np.max(np.abs(np.log(pri[0][acts].astype(np.float64)) - np.log(want_p))) <= 1e-4, (B, algo)
            assert abs(-rw[0] - want_v) <= 1e-4, (B, algo)
            occupied = [a for a in range(B * B) if a not in acts]
            assert not pri[0][occupied].any()
            for g, e in enumerate(envs):  # priors exist for the legal moves only: the legal share of the softmax
                illegal = sorted(set(range(B * B)) - set(e.leagel_actions()))
                assert not pri[g][illegal].any() and 0.0 < pri[g].astype(np.float64).sum() <= 1.0 + 1e-5
            assert np.array_equal(got['fused'][0].view(np.uint32), got['unfused'][0].view(np.uint32)), (B, algo)
            assert np.array_equal(got['fused'][1].view(np.uint64), got['unfused'][1].view(np.uint64)), (B, algo)
            # the trunk reading the leaf bitboards == the trunk reading the float planes the tree kernel writes
            assert np.array_equal(got['fused'][0].view(np.uint32), got['fused_planes'][0].view(np.uint32)), (B, algo)
            assert np.array_equal(got['fused'][1].view(np.uint64), got['fused_planes'][1].view(np.uint64)), (B, algo)
            if algo == 'split_f16':  # the tree kernel adding the GEMM's four K-quarter sums itself: the same bits
                if heads == 'auto':
                    reference_bits = (got['fused'][0].copy(), got['fused'][1].copy())
                else:
                    assert np.array_equal(got['fused'][0].view(np.uint32), reference_bits[0].view(np.uint32))
                    assert np.array_equal(got['fused'][1].view(np.uint64), reference_bits[1].view(np.uint64))
        evaluator.hip.close()


def test_puct_search_with_hip_net_vs_oracle():
    """Opt-in PUCT driven by the production evaluator (priors written by k_tree_step_raw are READ by this rule):
    the engine's tree == the oracle's PUCT restatement fed, leaf by leaf, the priors and value the device produces
    for that position (a one-game engine expanding that position as its root: the evaluation of a board does not
    depend on the batch it is in)."""
    import torch
    from rlzero_amd.engine import HipNetEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    for B, n, pre, sims, c in ((6, 4, [14, 15, 20], 120, 5.0), (9, 5, [40, 41], 100, 2.0)):
        torch.manual_seed(4)
        net = PolicyValueNet(B)
        with torch.no_grad():  # away from the near-uniform initial policy
            net.act_fc1.weight.mul_(20.0)
        net = net.to('cuda:0')
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=8)
        evaluator.deferred_priors = False   # (the one-game probe below follows the reference's rule: it must take the PUCT engine's route)
        probe = _engine(B, n, n_games=1, n_playout=2)

        def pvf(env, probe=probe, evaluator=evaluator):
            _set_roots(probe, [env], reset_trees=True)
            probe.sim_chunk(evaluator, 1)
            pri = probe.root_priors()[0]
            _, rw = probe.root_stats()
            legal = env.leagel_actions()
            return [(a, pri[a]) for a in legal], -float(rw[0])

        env = RefGomoku.from_moves(B, n, pre)
        eng = _engine(B, n, n_games=3, n_playout=sims, c_puct=c, score_mode='puct')
        _set_roots(eng, [env, RefGomoku(B, n), env], reset_trees=True)
        eng.simulate(evaluator, sims)
        eng.check()
        s = RefSearch(pvf, sims, c, score_mode='puct')
        acts, _ = s.simulate(env, 1.0)
        for g in (0, 2):
            visits, pri = eng.root_visits()[g], eng.root_priors()[g]
            assert [int(visits[a]) for a in acts] == [k.n for k in s.root.kids]
            assert [float(pri[a]) for a in acts] == [float(k.p) for k in s.root.kids]
            assert _hex_tree(eng.tree_dump(g)) == _hex_tree(tree_dump(s.root))
        assert max(k.n for k in s.root.kids) > 3 * max(1, min(k.n for k in s.root.kids))  # the priors shape the search
        eng.close()
        probe.close()
        evaluator.hip.close()


# ------------------------------------------------------------------ pure-MCTS opponent
def _device_stream_rand(seed, search):
    """rand(k) for the oracle that reproduces the device play-out's choices: a one-hot whose
    arg-max is rollout_pick(seed, game 0, sim, ply, k)."""
    from rlzero_amd.mcts.rollout_mcts import rollout_pick
    state = {'sim': -1, 'ply': 0}

    def rand(k):
        if search.sim_index != state['sim']:
            state['sim'], state['ply'] = search.sim_index, 0
        out = np.zeros(k)
        out[rollout_pick(seed, 0, state['sim'], state['ply'], k)] = 1.0
        state['ply'] += 1
        return out

    return rand


def test_rollout_search_vs_oracle():
    """Device random play-outs (k_eval_rollout) + the shared tree == the oracle's RolloutMCTS
    driven by the same per-(sim, ply) choices: identical trees and chosen moves."""
    from oracle.rollout_ref import RefRolloutSearch
    from rlzero_amd.games import GomokuEnv
    from rlzero_amd.mcts.rollout_mcts import RolloutMCTS
    for B, n, pre, sims, seed in ((3, 3, [], 60, 5), (3, 3, [4, 0, 2], 150, 6), (6, 4, [14, 15, 20, 21, 8], 200, 7),
                                  (9, 5, [40, 41, 31], 120, 8), (15, 5, [112, 113], 100, 9),
                                  (3, 3, [0, 1, 2, 4, 3, 5, 7], 30, 10)):
        env = GomokuEnv(B, n)
        env.reset()
        for m in pre:
            env.step(m)
        mcts = RolloutMCTS(n_playout=sims, c_puct=5)
        mcts.seed = seed
        move = mcts.simulate(env)
        ref = RefRolloutSearch(sims, 5)
        ref.rand = _device_stream_rand(seed, ref)
        want = ref.simulate(RefGomoku.from_moves(B, n, pre))
        assert move == want
        assert _hex_tree(mcts._engine.tree_dump(0)) == _hex_tree(tree_dump(ref.root))
        mcts._engine.close()


def test_rollout_player_game_through_reference_api():
    """policy_evaluate's pairing (tools/train_alphazero.py:139-163): AlphaZeroPlayer vs
    RolloutPlayer through GameControl.start_play runs to the end and is reproducible."""
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.mcts import AlphaZeroPlayer
    from rlzero_amd.mcts.rollout_mcts import RolloutPlayer
    results = []
    for _ in range(2):
        np.random.seed(123)
        env = GomokuEnv(6, 4)
        az = AlphaZeroPlayer(ev.vlin, n_playout=60, c_puct=5)
        pure = RolloutPlayer(n_playout=80, c_puct=5)
        winner = GameControl(env).start_play(az, pure, start_player=0, is_shown=0)
        results.append((winner, list(env.states.keys())))
        assert winner in (-1, 0, 1) and env.game_end_winner()[0]
        assert pure.mcts._root.explore_count == 0  # tree reset after every move
        az.mcts._engine.close()
        pure.mcts._engine.close()
    assert results[0] == results[1]


def test_batched_evaluation_equals_single_games(monkeypatch):
    """rlzero_amd.evaluate.BatchedEvaluation (policy_evaluate's games in lock-step, tools/train_alphazero.py:139-162): every game equals
    GameControl.start_play(AlphaZeroPlayer, RolloutPlayer) (game.py:61-94) played alone with the same draws -- the network player's
    two np.random.choice calls per move (alphazero_mcts.py:148,157) fed the batch's uniforms, the pure-MCTS player's play-out seed set to
    the batch's -- move for move, winner included; with the network seated second in some games both engines search at the same ply."""
    import torch
    from rlzero_amd.evaluate import BatchedEvaluation, rollout_seed
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts import AlphaZeroPlayer
    from rlzero_amd.mcts.rollout_mcts import RolloutPlayer
    from rlzero_amd.selfplay import draw_move, move_uniform
    B, n, seed, sims, pure_sims = 6, 4, 77, 40, 60
    torch.manual_seed(3)
    agent = AlphaZeroAgent(B, device='cuda:0')
    seats = [True, True, False, True, False, False]
    for use_graph in (False, True):
        duel = BatchedEvaluation.for_network(agent.policy_value_net, B, n, n_games=len(seats), n_playout=sims,
                                             rollout_playouts=pure_sims, seed=seed, use_graph=use_graph)
        results = duel.run(net_first=seats)
        again = duel.run(net_first=seats)  # the engines are reusable and the games reproducible
        assert [(r.moves, r.winner) for r in again] == [(r.moves, r.winner) for r in results]
        duel.close()
        assert all(r.winner in (-1, 0, 1) and len(r.moves) >= 2 * n - 1 for r in results)
        assert len({tuple(r.moves) for r in results}) > 1
        if use_graph:
            assert [(r.moves, r.winner) for r in results] == eager
        eager = [(r.moves, r.winner) for r in results]
    for g, net_first in enumerate(seats):
        env = GomokuEnv(B, n)
        calls = {}

        def choice(acts, p=None, g=g, env=env, calls=calls):
            ply = len(env.states)
            k = calls[ply] = calls.get(ply, -1) + 1   # 0: the draw that is dropped, 1: the one that is played
            return draw_move(acts, p, float(move_uniform(seed, g, 2 * ply + k)))

        monkeypatch.setattr(np.random, 'choice', choice)
        az = AlphaZeroPlayer(agent.policy_value_fn, n_playout=sims, c_puct=5)
        pure = RolloutPlayer(n_playout=pure_sims, c_puct=5)
        plain_get_action = pure.get_action

        def seeded(game_env, pure=pure, g=g, plain=plain_get_action, **kw):
            pure.mcts.seed = rollout_seed(seed, len(game_env.states)) ^ g
            return plain(game_env, **kw)

        pure.get_action = seeded
        winner = GameControl(env).start_play(az, pure, is_shown=0) if net_first else \
            GameControl(env).start_play(pure, az, is_shown=0)
        assert (list(env.states.keys()), winner) == (results[g].moves, results[g].winner), g
        assert sorted(calls.values()) == [1] * len(calls)   # two draws at every network ply
        assert results[g].net_won == (winner == (0 if net_first else 1))
        az.mcts._engine.close()
        pure.mcts._engine.close()


def test_batched_evaluation_connect4():
    """The same duel on Connect4 (actions = columns, the stone drops): every game replays on the rules twin to the winner the engines
    report, and ends."""
    import torch
    from rlzero_amd.evaluate import BatchedEvaluation
    from rlzero_amd.games.connect4.connect4_env import Connect4Env
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(5)
    net = PolicyValueNet(6, 7, 7).to('cuda:0')
    duel = BatchedEvaluation.for_network(net, (6, 7), 4, n_games=8, n_playout=50, rollout_playouts=80, seed=9, game='connect4',
                                         net_shape=(6, 7, 7))
    results = duel.run(net_first=[i % 2 == 0 for i in range(8)])
    duel.close()
    for r in results:
        env = Connect4Env(6, 7, 4)
        for i, m in enumerate(r.moves):
            assert not env.game_end_winner()[0] and m in env.leagel_actions()
            env.step(m)
        end, winner = env.game_end_winner()
        assert end and winner == r.winner


# ------------------------------------------------------------------ opt-in PUCT mode, noise
def _skewed(env):
    """Evaluator with non-uniform float32 priors (exactly the numbers the engine stores) and the
    vlin value."""
    legal = env.leagel_actions()
    raw = np.array([1 + (7 * a + 3) % 5 for a in legal], dtype=np.float32)
    probs = raw / np.float32(raw.sum())
    return list(zip(legal, probs)), ev.vlin_value(env.states, env.current_player())


def test_puct_mode_vs_oracle():
    """RZ_SCORE_PUCT (every child initialised, prior-weighted exploration, fp64, first maximum)
    against the oracle's restatement of node.py:105-117; tree reuse included."""
    from rlzero_amd.engine import HostEvaluator
    for B, n, pre, sims, c in ((3, 3, [], 80, 5.0), (6, 4, [14, 15, 20], 300, 5.0), (6, 4, [], 250, 1.5),
                               (9, 5, [40, 41, 31, 49], 300, 3.0)):
        env = RefGomoku.from_moves(B, n, pre)
        eng = _engine(B, n, n_games=1, n_playout=sims, c_puct=c, score_mode='puct')
        _set_roots(eng, [env], reset_trees=True)
        host = HostEvaluator(_skewed, lambda s0, s1, tm, last, B=B, n=n: _make_env(B, n, s0, s1, tm, last))
        eng.simulate(host, sims)
        eng.check()
        s = RefSearch(_skewed, sims, c, score_mode='puct')
        acts, _ = s.simulate(env, 1.0)
        visits, wsum, pri = eng.root_visits()[0], eng.root_wsum()[0], eng.root_priors()[0]
        assert [int(visits[a]) for a in acts] == [k.n for k in s.root.kids]
        assert [float(wsum[a]).hex() for a in acts] == [float(k.w).hex() for k in s.root.kids]
        assert [float(pri[a]) for a in acts] == [float(k.p) for k in s.root.kids]
        assert _hex_tree(eng.tree_dump(0)) == _hex_tree(tree_dump(s.root))
        assert max(k.n for k in s.root.kids) > 2 * min(k.n for k in s.root.kids)  # priors matter here
        # keep the most visited child's subtree and search again (update_with_move)
        best = acts[int(np.argmax([k.n for k in s.root.kids]))]
        eng.advance([best])
        eng.step([best])
        s.update_with_move(best)
        env.step(best)
        if not env.game_end_winner()[0]:
            eng.simulate(host, sims)
            s.simulate(env, 1.0)
            assert _hex_tree(eng.tree_dump(0)) == _hex_tree(tree_dump(s.root))
        eng.check()
        eng.close()


def test_dirichlet_noise_on_priors():
    """add_noise: stored prior = 0.75 p + 0.25 eta, eta ~ Dirichlet(0.3) over the legal moves
    (node.py:63-69).  The device stream is not numpy's: check the distribution's moments, that it
    sums to one, is reproducible for a seed, and that UCT_REF search results do not depend on it."""
    from rlzero_amd.engine import SyntheticEvaluator
    G, sims = 256, 30
    eng = _engine(9, 5, n_games=G, n_playout=sims, add_noise=True, noise_seed=11)
    eng.reset_games()
    eng.simulate(SyntheticEvaluator('v0'), sims)
    pri = eng.root_priors().astype(np.float64)
    vis = eng.root_visits()
    eng.check()
    k = 81
    eta = (pri - 0.75 / k) / 0.25
    assert np.allclose(pri.sum(1), 1.0, atol=1e-5) and (eta > -1e-6).all()
    assert abs(eta.mean() - 1.0 / k) < 1e-6
    want_var = (1.0 / k) * (1 - 1.0 / k) / (0.3 * k + 1)  # Dirichlet(alpha) marginal variance
    assert 0.8 * want_var < eta.var() < 1.25 * want_var
    # sparse like Dirichlet(0.3): most of the mass sits on few moves
    top = np.sort(eta, axis=1)[:, -8:].sum(1).mean()
    assert 0.35 < top < 0.75
    eng2 = _engine(9, 5, n_games=G, n_playout=sims, add_noise=True, noise_seed=11)
    eng2.reset_games()
    eng2.simulate(SyntheticEvaluator('v0'), sims)
    assert np.array_equal(eng2.root_priors(), eng.root_priors())
    eng3 = _engine(9, 5, n_games=G, n_playout=sims)
    eng3.reset_games()
    eng3.simulate(SyntheticEvaluator('v0'), sims)
    assert np.array_equal(eng3.root_visits(), vis)  # the reference's selection never reads the prior
    assert not np.array_equal(eng3.root_priors(), eng.root_priors())
    # rz_set_noise_keys: a game's stream follows its KEY, not its slot -- the keys of slots 0 .. G-1 reversed give the reversed
    # noise; masked slots keep theirs; keys = None restores the per-slot default (counters restart, so the first search again)
    keys = np.arange(G, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(5)
    outs = []
    for k_ in (keys, keys[::-1].copy()):
        eng2.reset_games()
        eng2.set_noise_keys(k_)
        eng2.simulate(SyntheticEvaluator('v0'), sims)
        outs.append(eng2.root_priors().copy())
    assert np.array_equal(outs[0], outs[1][::-1]) and not np.array_equal(outs[0], pri.astype(np.float32))
    eng2.reset_games()
    half = np.arange(G) < G // 2
    eng2.set_noise_keys(keys, mask=half)      # the upper half keeps the reversed keys -- and its running counters
    eng2.simulate(SyntheticEvaluator('v0'), sims)
    mixed = eng2.root_priors()
    assert np.array_equal(mixed[:G // 2], outs[0][:G // 2]) and not np.array_equal(mixed[G // 2:], outs[1][G // 2:])
    eng2.reset_games()
    eng2.set_noise_keys(None)
    eng2.simulate(SyntheticEvaluator('v0'), sims)
    assert np.array_equal(eng2.root_priors(), eng.root_priors())
    eng2.check()
    for e in (eng, eng2, eng3):
        e.close()


def test_compact_grid_trunk_is_the_same_kernel_on_less_lds(monkeypatch):
    """Boards of up to 7 columns in launches of more boards than HALF the CUs run k_trunk_split on a compact LDS grid (9 x 15 positions,
    67 KB: two workgroups per CU instead of one -- more boards than CUs in one round, two lanes' launches on the chip together;
    RZ_NET_COMPACT=0 keeps the 18 x 18 grid): the same instructions on other addresses -- log-probabilities and values bit for bit,
    for both channel-split shapes (one tile: 3 x 3, 5 x 5; two tiles: 6 x 6, 6 x 7, 7 x 7), above the CU count and just below it."""
    import torch
    from rlzero_amd.engine import HipNet
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    for shape in ((6, 7, 7), (6, 6, 36), (3, 3, 9), (7, 7, 49), (5, 5, 25)):
        rows, cols, acts = shape
        torch.manual_seed(rows * 10 + cols)
        net = PolicyValueNet(rows, cols, acts)
        for n in (2 * n_cus + 37, n_cus - 3):
            obs = (torch.rand(n, 4, rows, cols) > 0.5).float().to('cuda:0')
            out = {}
            for compact in ('0', '1'):
                monkeypatch.setenv('RZ_NET_COMPACT', compact)
                hip = HipNet(shape, 'cuda:0', max_boards=n).load_state_dict(net.state_dict())
                hip.set_heads_algo('parts')
                lp, v = hip.forward(obs)
                out[compact] = (lp.cpu().numpy().copy(), v.cpu().numpy().copy())
                hip.check_flags()
                hip.close()
            assert np.array_equal(out['0'][0].view(np.uint32), out['1'][0].view(np.uint32)), (shape, n)
            assert np.array_equal(out['0'][1].view(np.uint32), out['1'][1].view(np.uint32)), (shape, n)
            with torch.no_grad():
                lp64, v64 = net.double()(obs.cpu().double())
            net.float()
            assert np.max(np.abs(out['1'][0] - lp64.numpy())) <= 1e-4 and np.max(np.abs(out['1'][1] - v64.numpy()[:, 0])) <= 1e-4
