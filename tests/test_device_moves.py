"""The move step on the device (include/rlzero_hip.h: rz_play_*; BatchedSelfPlay.run_device): between two searches the host is
not in the loop -- the root visits are logged, the move of alphazero_mcts.py:147-148 is drawn, update_with_move (:96-103), env.step
+ game_end_winner (game.py:109-118), reset_player (:128) and the start of the next game all run as kernels enqueued moves ahead of
the host, which reads the log behind the GPU, forms pi with the reference's numpy expression and verifies every move.

Pinned here: the games are the oracle's (synthetic evaluator: bit-exact trees, moves, pi to 1e-12, z), they are bit for bit the
games of the host-driven loop (real net, lanes, hipGraphs, the resident search, Connect4, refills), and a draw that falls close to
an interval edge is decided by the host (forced here by a wide stall margin) without changing a single game."""
import os

import numpy as np
import pytest

from oracle import evaluators as ev
from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import RefPlayer, inverse_cdf_choice, self_play_game

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert [t.game_id for t in a] == [t.game_id for t in b]
    for x, y in zip(a, b):
        assert (x.winner, x.moves) == (y.winner, y.moves), x.game_id
        assert np.array_equal(np.asarray(x.pis).view(np.uint64), np.asarray(y.pis).view(np.uint64)), x.game_id


def test_device_moves_vs_oracle():
    """Synthetic evaluator, 8 slots, 24 games (slots refill themselves on the device): game by game the oracle's self-play game
    for the same uniforms -- moves, winner, pi, z, observation planes."""
    from rlzero_amd.engine import MCTSEngine, SyntheticEvaluator
    from rlzero_amd.selfplay import BatchedSelfPlay, move_uniform
    eng = MCTSEngine(6, 4, n_games=8, n_playout=60, device='cuda:0')
    sp = BatchedSelfPlay(eng, SyntheticEvaluator('vlin'), temperature=1.0, seed=5)
    trajs = sp.run_device(range(24))
    assert [t.game_id for t in trajs] == list(range(24))
    assert sp.sims_done == 60 * sum(len(t.moves) for t in trajs) and sp.moves_done == sum(len(t.moves) for t in trajs)
    for t in trajs[::2]:
        us = move_uniform(5, np.full(64, t.game_id), np.arange(64))
        player = RefPlayer(ev.vlin, 60, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
        winner, data, moves = self_play_game(RefGomoku(6, 4), player, temperature=1.0)
        assert (winner, moves) == (t.winner, t.moves)
        _, data2 = t.as_reference_tuple()
        for (s1, p1, z1), (s2, p2, z2) in zip(data, data2):
            assert (s1 == s2).all() and np.max(np.abs(p1 - p2)) <= 1e-12 and z1 == z2
    # the same object again, other games, and the host-driven loop on it afterwards: the slots are handed back clean
    more = sp.run_device(range(100, 110))
    host = sp.run(range(100, 110))
    _same(more, host)
    gid, ply, state, steps = eng.play_state()
    assert steps > 0
    eng.close()


@pytest.mark.parametrize('layout', ['resident_1lane', 'graphs_2lanes', 'graphs_3lanes_9x9', 'connect4_2lanes'])
def test_device_moves_equal_host_moves(layout):
    """The production evaluator on the production routes: run_device() == run() bit for bit (moves, pi, winners), with more
    games than slots."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(2)
    if layout == 'connect4_2lanes':
        net = PolicyValueNet(6, 7, 7).to('cuda:0')
        kw = dict(board=(6, 7), n_in_row=4, n_games=10, n_playout=50, game='connect4', net_shape=(6, 7, 7), lanes=2,
                  use_graph=True, sims_per_graph=8, resident_search=False)
    elif layout == 'graphs_3lanes_9x9':
        net = PolicyValueNet(9).to('cuda:0')
        kw = dict(board=9, n_in_row=5, n_games=9, n_playout=40, lanes=3, use_graph=True, sims_per_graph=8, resident_search=False)
    elif layout == 'graphs_2lanes':
        net = PolicyValueNet(11).to('cuda:0')
        kw = dict(board=11, n_in_row=5, n_games=8, n_playout=48, lanes=2, use_graph=True, sims_per_graph=16, resident_search=False)
    else:
        net = PolicyValueNet(6).to('cuda:0')
        kw = dict(board=6, n_in_row=4, n_games=6, n_playout=40, lanes=1)
    ids = list(range(3, 3 + 2 * kw['n_games'] + 3))
    host = BatchedSelfPlay.for_network(net, device='cuda:0', temperature=1.0, seed=9, **kw)
    a = host.run(ids, pipelined=len(host.lanes) > 1)
    dev = BatchedSelfPlay.for_network(net, device='cuda:0', temperature=1.0, seed=9, **kw)
    b = dev.run_device(ids)
    _same(a, b)
    assert dev.sims_done == host.sims_done and dev.moves_done == host.moves_done
    for sp in (host, dev):
        for st in sp.check():
            assert st.reuse_dropped == 0
        for lane in sp.lanes:
            lane.evaluator.hip.check_flags()
            lane.eng.close()


@pytest.mark.parametrize('temperature', [1e-3, 0.5])
def test_device_moves_at_other_temperatures(temperature):
    """temperature = 1e-3 is get_action's default (alphazero_mcts.py:136): pi is all but one-hot, exp() underflows for every other
    child and equal maxima share the probability -- the device's fp64 exp / log draw (or its stall, the margin scales with 1 / T)
    gives the host loop's games and the oracle's, bit for bit; 0.5 for a temperature between."""
    from rlzero_amd.engine import MCTSEngine, SyntheticEvaluator
    from rlzero_amd.selfplay import BatchedSelfPlay, move_uniform
    eng = MCTSEngine(6, 4, n_games=6, n_playout=50, device='cuda:0')
    sp = BatchedSelfPlay(eng, SyntheticEvaluator('vlin'), temperature=temperature, seed=8)
    dev = sp.run_device(range(14))
    host = sp.run(range(14))
    _same(dev, host)
    for t in dev[::3]:
        us = move_uniform(8, np.full(64, t.game_id), np.arange(64))
        player = RefPlayer(ev.vlin, 50, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
        winner, data, moves = self_play_game(RefGomoku(6, 4), player, temperature=temperature)
        assert (winner, moves) == (t.winner, t.moves)
        for (_, p1, _), p2 in zip(data, t.pis):
            assert np.max(np.abs(p1 - p2)) <= 1e-12
    eng.close()


def test_a_draw_near_an_interval_edge_is_the_host_s():
    """stall_margin = 0.08: roughly one draw in six lies that close to an edge of its interval -- the device does not draw, the
    slot sits out the coming searches, the host decides with numpy and hands the move back.  The games are those of the default
    margin (where no draw stalls), bit for bit; a stalled slot's search is not repeated (simulations counted once)."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(4)
    net = PolicyValueNet(6).to('cuda:0')
    kw = dict(board=6, n_in_row=4, n_games=7, n_playout=40, device='cuda:0', temperature=1.0, seed=21, lanes=2,
              use_graph=True, sims_per_graph=8, resident_search=False)
    ids = list(range(20))
    plain = BatchedSelfPlay.for_network(net, **kw)
    a = plain.run_device(ids)
    assert plain.stalls_resolved == 0
    wide = BatchedSelfPlay.for_network(net, **kw)
    wide.device_attach(queue_capacity=64, stall_margin=0.08)
    b = wide.run_device(ids)
    _same(a, b)
    assert wide.stalls_resolved >= 5 and wide.sims_done == plain.sims_done
    # (the margin is reported per record: every un-stalled draw of the wide run lay farther than 0.08 from its edges)
    for sp in (plain, wide):
        sp.check()
        for lane in sp.lanes:
            lane.eng.close()


def test_device_moves_with_a_host_callable_and_the_modes_that_refuse():
    """Any `policy_value_fn` callable (evaluated per leaf on the host, alphazero_mcts.py:28-31,59) can sit in front of the device's
    move step: the searches are host-paced, the moves still drawn, applied and logged by the device -- the oracle's games.  The
    opt-in mode with several simulations in flight is refused (the move step assumes the reference's one simulation per tree)."""
    from rlzero_amd._hip import HipError
    from rlzero_amd.engine import HostEvaluator, MCTSEngine, SyntheticEvaluator
    from rlzero_amd.games.gomoku.gomoku_env import GomokuEnv
    from rlzero_amd.selfplay import BatchedSelfPlay, move_uniform
    eng = MCTSEngine(3, 3, n_games=3, n_playout=30, device='cuda:0')
    host = HostEvaluator(ev.vlin, lambda s0, s1, to_move, last: GomokuEnv.from_bitboards(3, 3, s0, s1, to_move, last))
    sp = BatchedSelfPlay(eng, host, temperature=1.0, seed=2)
    trajs = sp.run_device(range(5))
    assert [t.game_id for t in trajs] == list(range(5))
    for t in trajs:
        us = move_uniform(2, np.full(16, t.game_id), np.arange(16))
        player = RefPlayer(ev.vlin, 30, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
        winner, data, moves = self_play_game(RefGomoku(3, 3), player, temperature=1.0)
        assert (winner, moves) == (t.winner, t.moves)
        for (_, p1, _), p2 in zip(data, t.pis):
            assert np.max(np.abs(p1 - p2)) <= 1e-12
    eng.close()
    eng2 = MCTSEngine(6, 4, n_games=4, n_playout=32, device='cuda:0', sims_in_flight=4)
    sp2 = BatchedSelfPlay(eng2, SyntheticEvaluator('vlin'), temperature=1.0, seed=2)
    with pytest.raises(HipError, match='one simulation in flight'):
        sp2.run_device(range(4))
    eng2.close()


def test_lanes_by_measurement_on_the_gpu():
    """`for_network(lanes='measure')` (what a chip the tables were not measured on gets by itself): the table's pick and its
    neighbours are built, timed for two moves with the device-driven loop and closed; the layout kept plays the games every layout
    plays."""
    import torch
    from rlzero_amd import selfplay as sp_mod
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay
    torch.manual_seed(6)
    net = PolicyValueNet(6).to('cuda:0')
    sp_mod._LANE_CACHE.clear()
    kw = dict(board=6, n_in_row=4, n_games=12, n_playout=24, device='cuda:0', temperature=1.0, seed=3, use_graph=True, sims_per_graph=8)
    sp = BatchedSelfPlay.for_network(net, lanes='measure', **kw)
    assert sp.lanes_measured is not None and len(sp.lanes) in sp.lanes_measured and len(sp.lanes_measured) >= 2
    assert all(rate > 0 for rate in sp.lanes_measured.values()) and len(sp_mod._LANE_CACHE) == 1
    again = BatchedSelfPlay.for_network(net, lanes='measure', **kw)   # (cached: nothing is timed again)
    assert len(again.lanes) == len(sp.lanes) and again.lanes_measured == sp.lanes_measured
    one = BatchedSelfPlay.for_network(net, lanes=1, **kw)
    _same(sp.run_device(range(20)), one.run(range(20)))
    for s_ in (sp, again, one):
        for lane in s_.lanes:
            lane.eng.close()


@pytest.mark.gpu
def test_the_baseline_batch_does_not_depend_on_import_order():
    """512 games at 15x15 / 800 are ONE resident lane (k_delta_res, two games per CU) on one stream: a process that touched the GPU
    BEFORE importing rlzero_amd -- the runtime then has its default four hardware queues, where round 5's four-lane layout fell from
    10.2 to 6.1 M simulations / s -- gets the same layout and, within 10 %, the rate of a process that imported the package first."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
os.environ.pop('GPU_MAX_HW_QUEUES', None)
late = sys.argv[1] == 'late'
import torch
if late:
    torch.zeros(8, device='cuda:0').sum().item()     # the HIP runtime starts here, with its default hardware queues
import rlzero_amd
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
from rlzero_amd.selfplay import BatchedSelfPlay, time_moves
torch.manual_seed(0)
net = PolicyValueNet(15).to('cuda:0')
sp = BatchedSelfPlay.for_network(net, 15, 5, n_games=512, n_playout=800, seed=1)
lane = sp.lanes[0]
out = {'late': late, 'hw_queues': rlzero_amd.HW_QUEUES, 'too_late': rlzero_amd.HW_QUEUES_TOO_LATE, 'lanes': len(sp.lanes),
       'resident': bool(lane.evaluator.resident_ok(lane.eng) and lane.evaluator.resident_delta_ok(lane.eng)), 'rate': time_moves(sp, moves=3)}
print('RESULT ' + json.dumps(out))
'''
    results = {}
    for mode in ('early', 'late'):
        run = subprocess.run([sys.executable, '-c', code, mode], capture_output=True, text=True, timeout=600,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith('RESULT ')]
        assert run.returncode == 0 and lines, (mode, run.stderr[-2000:])
        results[mode] = json.loads(lines[-1][7:])
    early, late = results['early'], results['late']
    assert late['too_late'] and not early['too_late'] and early['hw_queues'] >= 8 and late['hw_queues'] < 8
    assert early['lanes'] == late['lanes'] == 1 and early['resident'] and late['resident']
    assert late['rate'] >= 0.9 * early['rate'] and early['rate'] > 1.2e7, results
