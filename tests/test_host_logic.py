"""Host-side mirror of the reference interface (no GPU): GomokuEnv rules and observation
against the golden fixtures, GameControl bookkeeping with a scripted player."""
import copy
import os

import numpy as np
import pytest
from conftest import REPO, bits_of_planes

from rlzero_amd.games import BaseEnv, Error, Game, GameControl, GomokuEnv
from rlzero_amd.mcts import AlphaZeroPlayer, MCTSPlayer, Player


def _replay(case):
    env = GomokuEnv(case['B'], case['n'])
    env.reset()
    for ply in case['plies']:
        obs, reward, win, _ = env.step(ply['a'])
        ended, winner = env.game_end_winner()
        won, who = env.has_a_winner()
        assert (ended, winner, won, who) == (ply['ended'], ply['winner'], ply['won'], ply['who'])
        assert win == ply['won']
        if won:
            assert reward == 1  # the mover wins
        assert len(env.leagel_actions()) == ply['n_legal']
        assert env.current_player() == ply['to_move'] and env.last_move == ply['last']
        assert bits_of_planes(obs) == ply['obs']
        assert bits_of_planes(env.current_state()) == ply['obs']
        s0, s1 = env.bitboards()
        twin = GomokuEnv.from_bitboards(case['B'], case['n'], s0, s1, env.current_player(), env.last_move)
        assert bits_of_planes(twin.current_state()) == ply['obs']
        assert twin.game_end_winner() == (ended, winner)
        assert twin.leagel_actions() == env.leagel_actions() and twin.states == env.states


def test_env_rules_random(g1):
    for case in g1['random']:
        _replay(case)


def test_env_rules_handmade(g1):
    for case in g1['handmade']:
        _replay(case)


def test_env_interface_and_errors():
    assert Game is BaseEnv and MCTSPlayer is AlphaZeroPlayer
    env = GomokuEnv(board_size=6, n_in_row=4)
    obs = env.reset()
    assert obs.shape == (4, 6, 6) and obs.dtype == np.float64
    assert env.players == [0, 1] and env.last_move == -1 and env.states == {}
    assert env.leagel_actions() == list(range(36)) and env.legal_actions(0) is env.leagel_actions()
    with pytest.raises(AssertionError):
        env.step(36)
    env.step(7)
    with pytest.raises(AssertionError):
        env.step(7)
    assert env.move_to_location(7) == [1, 1] and env.location_to_move([1, 1]) == 7
    assert env.location_to_move([6, 0]) == -1 and env.location_to_move([1]) == -1
    with pytest.raises(Error):
        GomokuEnv(3, 5).reset()
    with pytest.raises(Error):
        env.reset(start_player_idx=2)
    env.reset(start_player_idx=1)
    assert env.current_player() == 1
    twin = copy.deepcopy(env)
    twin.step(0)
    assert env.states == {} and twin.states == {0: 1}
    assert env.clone().states == {}


class Scripted(Player):

    def __init__(self, moves):
        super().__init__()
        self.moves = list(moves)
        self.resets = 0

    def get_action(self, env, temperature=1e-3, return_prob=False):
        move = self.moves.pop(0)
        pi = np.zeros(env.board_size ** 2)
        pi[move] = 1.0
        return (move, pi) if return_prob else move

    def reset_player(self):
        self.resets += 1


def test_game_control_self_play_bookkeeping():
    game = GameControl(GomokuEnv(3, 3))
    player = Scripted([0, 3, 1, 4, 2])  # player 0 wins with 0,1,2
    winner, data = game.start_self_play(player, temperature=1.0)
    data = list(data)
    assert winner == 0 and player.resets == 1 and len(data) == 5
    assert [z for _, _, z in data] == [1.0, -1.0, 1.0, -1.0, 1.0]
    assert data[0][0].shape == (4, 3, 3) and data[0][0][3].all() and not data[1][0][3].any()
    player = Scripted([0, 1, 2, 4, 3, 5, 7, 6, 8])
    winner, data = game.start_self_play(player)
    assert winner == -1 and [z for _, _, z in data] == [0.0] * 9


def test_game_control_two_players():
    game = GameControl(GomokuEnv(3, 3))
    p1, p2 = Scripted([4, 0, 8]), Scripted([1, 2])
    with pytest.raises(Error):
        game.start_play(p1, p2, start_player=2, is_shown=0)
    # 4,1,0,2,8 -> player 0 completes the diagonal 0,4,8
    assert game.start_play(p1, p2, start_player=1, is_shown=0) == 0
    assert p1.get_player_id() == 0 and p2.get_player_id() == 1


def test_plan_lanes():
    """One lane up to a round of boards (one per CU); beyond, lanes with un-capped trunks and the LDS-free FC GEMM: three up to 1.75
    rounds, four up to 2.75 rounds where 8 hardware queues carry them (three where only HIP's default 4 do), two beyond."""
    from rlzero_amd.selfplay import plan_lanes
    assert plan_lanes(1) == (1, 0, 'auto') and plan_lanes(256) == (1, 0, 'auto')
    assert plan_lanes(320, hw_queues=8) == (3, 0, 'parts') and plan_lanes(447, hw_queues=8) == (3, 0, 'parts')
    # the 512 games per GPU of BASELINE.json configs[3]
    assert plan_lanes(448, hw_queues=8) == (4, 0, 'parts') and plan_lanes(512, hw_queues=8) == (4, 0, 'parts')
    assert plan_lanes(704, hw_queues=8) == (4, 0, 'parts') and plan_lanes(512, hw_queues=16) == (4, 0, 'parts')
    with pytest.warns(RuntimeWarning, match='four lanes'):
        assert plan_lanes(512, hw_queues=4) == (3, 0, 'parts')   # a fourth stream would share a hardware queue
    assert plan_lanes(768, hw_queues=8) == (2, 0, 'parts') and plan_lanes(1536, hw_queues=8) == (2, 0, 'parts')
    assert plan_lanes(100, n_cus=32, hw_queues=8) == (2, 0, 'parts')
    # the deferred-priors route (a step = trunk -> tree step): profiles/r04/lane_sweep.txt
    d = lambda n, q=8: plan_lanes(n, hw_queues=q, deferred=True)[0]  # noqa: E731
    assert [d(n) for n in (1, 128, 192, 193, 256, 257, 320, 384, 447, 448, 511, 512, 640, 704, 705, 768, 1536)] == \
        [1, 1, 2, 2, 2, 3, 3, 3, 3, 2, 2, 4, 4, 4, 2, 2, 2]
    assert d(512, 4) == 2 and d(640, 4) == 2 and plan_lanes(512, hw_queues=8, deferred=True) == (4, 0, 'parts')
    # small boards on the two-launch step (Connect4: profiles/r04/small_boards_lanes.txt): two lanes above one round of boards
    assert plan_lanes(512, hw_queues=8, cells=42) == (2, 0, 'parts') and plan_lanes(1024, hw_queues=8, cells=36) == (2, 0, 'parts')
    assert plan_lanes(256, hw_queues=8, cells=42) == (1, 0, 'auto') and plan_lanes(512, hw_queues=8, cells=81) == (4, 0, 'parts')
    # K simulations in flight on a small board (9x9, K = 16: profiles/r04/k_in_flight_lanes.txt): three lanes from two rounds of leaves, four from four
    assert plan_lanes(512, hw_queues=8, cells=81, in_flight=16)[0] == 3 and plan_lanes(1024, hw_queues=8, cells=81, in_flight=16)[0] == 4
    assert plan_lanes(1024, hw_queues=4, cells=81, in_flight=16)[0] == 3 and plan_lanes(256, hw_queues=8, cells=81, in_flight=16)[0] == 1
    assert plan_lanes(2048, hw_queues=8, cells=225, in_flight=16)[0] == 2   # (15x15 keeps the table)
    # the receptive-field trunk: the resident search holds two games per CU -- up to 2 x CUs games are one lane, whatever the queues
    r = lambda n, q=8: plan_lanes(n, hw_queues=q, deferred=True, resident_per_cu=2)   # noqa: E731
    assert r(512) == (1, 0, 'auto') and r(512, 4) == (1, 0, 'auto') and r(256) == (1, 0, 'auto') and r(1) == (1, 0, 'auto')
    assert r(513)[0] == 4 and r(704)[0] == 4 and r(767)[0] == 2 and plan_lanes(512, hw_queues=8, resident_per_cu=2)[0] == 4   # (between one and 1.5 rounds, and off the deferred route: the table)
    assert r(768)[0] == 1 and r(1024)[0] == 1 and r(1536)[0] == 1 and r(4096)[0] == 1   # (from 1.5 rounds on: one lane, the launch runs in rounds)
    assert plan_lanes(1536, hw_queues=8, deferred=True)[0] == 2   # (one resident workgroup per CU: no rounds)
    assert plan_lanes(128, n_cus=64, hw_queues=8, deferred=True, resident_per_cu=2)[0] == 1
    # boards of the compact LDS grid (Connect4, 6x6, 7x7): two resident games per CU -> one lane up to 2 x CUs games, the lanes beyond
    c = lambda n, cells=42, per_cu=2: plan_lanes(n, hw_queues=8, cells=cells, resident_per_cu=per_cu)   # noqa: E731
    assert c(512) == (1, 0, 'auto') and c(384)[0] == 1 and c(257)[0] == 1 and c(512, 49)[0] == 1 and c(512, 36)[0] == 1
    assert c(513)[0] == 2 and c(1024)[0] == 2 and c(512, per_cu=1)[0] == 2 and c(256)[0] == 1
    assert c(512, 81)[0] == 4   # (9x9 has no compact grid: the table)
    from rlzero_amd.engine import compact_grid_board
    assert all(compact_grid_board(r, k) for r, k in ((6, 7), (6, 6), (3, 3), (7, 7), (5, 5)))
    assert not any(compact_grid_board(r, k) for r, k in ((9, 9), (8, 8), (15, 15), (10, 10)))


def test_lanes_by_measurement_fallback():
    """Off the table's chip the layout is measured: the table's pick and its neighbours are built, timed (a fake timer here) and
    closed, the fastest wins (ties: fewer lanes), the answer is cached per key."""
    from rlzero_amd import selfplay as sp_mod
    assert sp_mod.lane_candidates(4, 8, 512) == [3, 4] and sp_mod.lane_candidates(2, 8, 512) == [1, 2, 3]
    assert sp_mod.lane_candidates(3, 4, 512) == [2, 3] and sp_mod.lane_candidates(1, 8, 512) == [1, 2] and sp_mod.lane_candidates(2, 8, 1) == [1]
    built, closed = [], []

    class FakeEngine(object):
        def __init__(self, tag):
            self.tag = tag

        def close(self):
            closed.append(self.tag)

    class FakeLane(object):
        def __init__(self, tag):
            self.eng = FakeEngine(tag)

    class FakeSp(object):
        def __init__(self, n):
            self.lanes = [FakeLane((n, i)) for i in range(n)]

    def build(n):
        built.append(n)
        return FakeSp(n)

    cache = {}
    rates = {1: 5.0, 2: 9.0, 3: 9.0, 4: 7.0}
    best, seen = sp_mod.choose_lanes_by_measurement(('gfx', 128, 15), 2, 8, 256, build, timer=lambda sp: rates[len(sp.lanes)], cache=cache)
    assert best == 2 and seen == {1: 5.0, 2: 9.0, 3: 9.0} and built == [1, 2, 3]      # the tie between 2 and 3 goes to fewer lanes
    assert sorted(closed) == [(1, 0), (2, 0), (2, 1), (3, 0), (3, 1), (3, 2)]          # every timed layout was closed
    again = sp_mod.choose_lanes_by_measurement(('gfx', 128, 15), 2, 8, 256, build, timer=lambda sp: 0.0, cache=cache)
    assert again == (2, seen) and built == [1, 2, 3]                                    # cached: nothing built again
    best, seen = sp_mod.choose_lanes_by_measurement(('gfx', 128, 9), 3, 8, 256, build, timer=lambda sp: rates[len(sp.lanes)], cache=cache)
    assert best == 2 and sorted(seen) == [2, 3, 4]

    def failing(sp):
        raise RuntimeError('timer failed')
    del closed[:]
    with pytest.raises(RuntimeError):
        sp_mod.choose_lanes_by_measurement(('other', ), 1, 8, 64, build, timer=failing, cache=cache)
    assert closed == [(1, 0)] and ('other', ) not in cache                              # closed even then, nothing cached
    # a candidate the device refuses (HipError: e.g. a mode the timer's move step does not serve) is no candidate; when none can be
    # timed the table's pick stands and nothing is cached as a measurement
    from rlzero_amd._hip import HipError
    del closed[:]

    def refusing(sp):
        if len(sp.lanes) == 2:
            raise HipError('refused')
        return rates[len(sp.lanes)]
    best, seen = sp_mod.choose_lanes_by_measurement(('k', 2), 2, 8, 256, build, timer=refusing, cache=cache)
    assert best == 3 and sorted(seen) == [1, 3] and sorted(closed) == [(1, 0), (2, 0), (2, 1), (3, 0), (3, 1), (3, 2)]

    def always(sp):
        raise HipError('refused')
    assert sp_mod.choose_lanes_by_measurement(('k', 3), 2, 8, 256, build, timer=always, cache=cache) == (2, None) and ('k', 3) not in cache
    assert sp_mod.TABLE_CUS == 256


def test_human_player_asks_until_the_move_is_legal(capsys):
    """rlzero/mcts/player.py:33-57: "row,col" from the console, asked again after nonsense, an occupied cell or a cell off the board."""
    from rlzero_amd.mcts import HumanPlayer
    env = GomokuEnv(3, 3)
    env.reset()
    env.step(4)
    answers = iter(['x', '1,1', '9,9', '2,0'])
    human = HumanPlayer(player_id=1, player_name='me', ask=lambda prompt: next(answers))
    assert human.can_click and human.get_action(env) == env.location_to_move([2, 0])
    assert capsys.readouterr().out.count('invalid move') == 3 and str(human) == 'HumanPlayer, id: 1, name me.'
    human.reset_player()


def test_hw_queues_are_claimed_on_import():
    """rlzero_amd sets GPU_MAX_HW_QUEUES before the HIP runtime starts (unless the caller chose a value) and remembers what holds."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import rlzero_amd; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), rlzero_amd.HW_QUEUES)" % REPO)
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    assert subprocess.check_output([sys.executable, '-c', code], env=env).decode().split() == ['8', '8']
    env['GPU_MAX_HW_QUEUES'] = '4'
    assert subprocess.check_output([sys.executable, '-c', code], env=env).decode().split() == ['4', '4']
    del env['GPU_MAX_HW_QUEUES']
    assert subprocess.check_output([sys.executable, '-c', code], env=dict(env, RZ_HW_QUEUES='6')).decode().split() == ['6', '6']
    # the explicit call, before the runtime starts: the number holds
    code = ("import os, sys; sys.path.insert(0, %r); import rlzero_amd; n = rlzero_amd.configure(hw_queues=12); "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), n, rlzero_amd.HW_QUEUES)" % REPO)
    assert subprocess.check_output([sys.executable, '-c', code], env=env).decode().split() == ['12', '12', '12']
    # a late import (the runtime of the process holds /dev/kfd already -- simulated: no GPU here): nothing is claimed, configure()
    # warns, and plan_lanes() warns when it falls back from four lanes to three
    code = """
import os, sys, warnings
sys.path.insert(0, %r)
import rlzero_amd as rz
rz._runtime_started = lambda: True
os.environ.pop('GPU_MAX_HW_QUEUES', None)
rz.HW_QUEUES, rz.HW_QUEUES_TOO_LATE = rz._claim_hw_queues()
print(rz.HW_QUEUES, rz.HW_QUEUES_TOO_LATE, os.environ.get('GPU_MAX_HW_QUEUES'))
from rlzero_amd.selfplay import plan_lanes
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    n = rz.configure(hw_queues=8)
    lanes = plan_lanes(512)[0]
print(n, lanes, [x.category.__name__ for x in w], 'after the HIP runtime' in str(w[-1].message))
""" % REPO
    out = subprocess.check_output([sys.executable, '-c', code], env=env).decode().splitlines()
    assert out[0].split() == ['4', 'True', 'None']
    assert out[1] == "4 3 ['RuntimeWarning', 'RuntimeWarning'] True"


def test_batched_pi_and_moves_bit_identical_to_per_game_expressions():
    """The vectorised host step of BatchedSelfPlay reproduces, bit for bit, the reference's per-game numpy
    expressions (softmax of log visits over the legal moves, numpy's inverse-CDF choice)."""
    from rlzero_amd.selfplay import batch_pi_and_moves, draw_move, visits_to_pi
    rng = np.random.default_rng(0)
    for temperature in (1.0, 1e-3, 0.5):
        for _ in range(6):
            R, A = 48, 225
            legal = rng.random((R, A)) < rng.random((R, 1))
            legal[:, 0] |= ~legal.any(axis=1)
            visits = (rng.integers(0, 50, size=(R, A)) * (rng.random((R, A)) < 0.6) * legal).astype(np.int32)
            us = rng.random(R)
            pi, mv = batch_pi_and_moves(visits, legal, temperature, us)
            for r in range(R):
                acts = np.nonzero(legal[r])[0]
                p = visits_to_pi(visits[r, acts], temperature)
                full = np.zeros(A)
                full[acts] = p
                assert np.array_equal(full, pi[r]) and draw_move(acts, p, us[r]) == mv[r]




def test_fc_in_trunk_rule():
    """The FC layers ride inside the trunk's launch only where their weights are a small stream per board (<= 40 KB of hi + lo
    f16, boards of up to 10 rows): TicTacToe, 6x6, Connect4 -- not 9x9 (146 KB: measured slower), never the 15x15 of the metric."""
    from rlzero_amd.selfplay import fc_in_trunk_pays
    assert fc_in_trunk_pays(3, 3, 9) and fc_in_trunk_pays(6, 6, 36) and fc_in_trunk_pays(6, 7, 7)
    assert not fc_in_trunk_pays(7, 7, 49) and not fc_in_trunk_pays(9, 9, 81) and not fc_in_trunk_pays(15, 15, 225)
    assert not fc_in_trunk_pays(11, 3, 33)  # a board of more than 10 rows has no room for the pieces in LDS


def test_bench_reads_the_committed_counter_runs():
    """bench.pmc_traffic: HBM-side bytes per launch from profiles/r03/pmc_traffic.json, one entry per workload on the line (the
    rule variants of a geometry keyed separately); None for a workload nobody profiled."""
    import bench
    head = 'gomoku15x15_n5_selfplay_800sims_per_move_512games_per_gpu'
    for kernel in ('k_trunk', 'k_heads', 'k_tree_step'):
        assert bench.pmc_traffic(kernel, head) > 1e6
        assert bench.pmc_traffic(kernel, 'gomoku15x15_n5_selfplay_800sims_per_move_1536games_per_gpu') > bench.pmc_traffic(kernel, head)
    assert bench.pmc_traffic('k_tree_step', head + '+puct') > bench.pmc_traffic('k_tree_step', head)   # every level scans all children
    assert bench.pmc_traffic('k_mz_search', 'muzero_cartpole_v1_50sims_per_move_8192envs_per_gpu') > 1e9
    assert bench.pmc_traffic('k_trunk', 'gomoku19x19_n5_selfplay_800sims_per_move_512games_per_gpu') is None
    # the trunk of a 256-board launch writes exactly the f16 feature pieces: 256 boards x 1350 features x (hi + lo) x 2 bytes
    rec = __import__('json').load(open(__import__('os').path.join(bench.REPO, 'profiles', 'r03', 'pmc_traffic.json')))
    # what a trunk launch writes is exactly its boards' head features as f16 pieces: 128 boards (a lane of the headline) x 1350 x 4 B
    assert abs(rec[head]['kernels']['k_trunk']['write_size_kb'] * 1024 - 128 * 1350 * 4) < 0.03 * 128 * 1350 * 4


def test_trajectory_payload_round_trip():
    """The byte payload of gather_trajectories (header | moves | pi in one buffer) cuts up again into the same trajectories, for both
    pi widths, ragged games and an empty list."""
    from rlzero_amd.selfplay import Trajectory, _payload_bytes, _payload_split, pack_trajectories, unpack_trajectories
    rng = np.random.default_rng(3)
    trajs = []
    for gid, plies in ((5, 9), (2, 1), (11, 4)):
        pis = rng.random((plies, 9))
        pis /= pis.sum(axis=1, keepdims=True)
        trajs.append(Trajectory(gid, 3, 3, rng.integers(0, 9, plies), pis, int(rng.integers(-1, 2))))
    for dtype in (np.float64, np.float32):
        for batch in (trajs, []):
            header, moves, pis = pack_trajectories(batch, 9)
            raw = _payload_bytes(header, moves, pis, dtype)
            assert raw.dtype == np.uint8 and raw.size == 32 * len(batch) + moves.size * (8 + 9 * np.dtype(dtype).itemsize)
            back = unpack_trajectories(*_payload_split(raw, len(batch), moves.size, 9, dtype), 3, 3)
            assert [(t.game_id, t.moves, t.winner) for t in back] == [(t.game_id, t.moves, t.winner) for t in batch]
            for a, b in zip(back, batch):
                assert np.array_equal(a.pis, b.pis.astype(dtype).astype(np.float64))


def test_fp8_model_rounds_to_the_ocp_grids():
    """oracle/fp8_cross_ref.py (the float64 model of the opt-in FP8 cross terms): e4m3fn and e5m2 roundings -- every value on the
    grid is a fixed point, midpoints go to the even neighbour, subnormals keep the smallest normal's step, saturation at 448 / 57344 --
    and the three-term split it models reproduces a float32 product to 2^-21."""
    import torch
    from oracle import fp8_cross_ref as m
    grid4 = torch.tensor([s * (1 + k / 8.0) * 2.0 ** e for s in (1, -1) for e in range(-6, 9) for k in range(8) if (1 + k / 8.0) * 2.0 ** e <= 448] +
                         [k * 2.0 ** -9 for k in range(8)], dtype=torch.float64)
    assert torch.equal(m.e4m3(grid4), grid4)
    grid5 = torch.tensor([s * (1 + k / 4.0) * 2.0 ** e for s in (1, -1) for e in range(-14, 16) for k in range(4)] + [k * 2.0 ** -16 for k in range(4)],
                         dtype=torch.float64)
    assert torch.equal(m.e5m2(grid5), grid5)
    x = torch.tensor([1.0625, 1.1875, 1.0626, 500.0, -1000.0, 2.0 ** -10, 1.5 * 2.0 ** -10, 0.9 * 2.0 ** -10], dtype=torch.float64)
    assert m.e4m3(x).tolist() == [1.0, 1.25, 1.125, 448.0, -448.0, 0.0, 2.0 ** -9, 0.0]
    y = torch.tensor([1.125, 1.375, 1.126, 60000.0, 61440.0, 1e9, 2.0 ** -17, 1.5 * 2.0 ** -16], dtype=torch.float64)
    assert m.e5m2(y).tolist() == [1.0, 1.5, 1.25, 57344.0, 57344.0, 57344.0, 0.0, 2.0 ** -15]
    torch.manual_seed(0)
    a, w = torch.randn(1000).float(), torch.randn(1000).float()
    ah, al = m.split(a)
    wh, wl = m.split(w)
    assert float(((ah * wh + ah * wl + al * wh) - a.double() * w.double()).abs().max()) < 2.0 ** -21 * 16
