#!/usr/bin/env python3
"""Round 6 soak: the resident search with the receptive-field trunk (k_delta_res, two games per CU, rounds) against the two-launch step
with the FULL-BOARD trunk (k_trunk_rows + k_tree_step_def), on every board size of the row kernel, from random positions between
the empty and the nearly full board (deep paths: more than four changed cells -> the route without a base; windows beyond the record
budget), three moves with tree reuse, noise.  Root visits of every game and whole trees of a sample must be equal bit for bit.
    python3 profiles/soak_r06.py > profiles/r06/soak.txt        (imports the oracle's rules only to make positions: test infrastructure)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import test_deferred as T   # noqa: E402  (helpers: _positions, _set_roots, _net, _whole_tree)
from rlzero_amd.engine import HipNetEvaluator, MCTSEngine   # noqa: E402

from oracle.gomoku_ref import RefGomoku   # noqa: E402


def deep_positions(B, count, seed):
    """Non-terminal positions with 5 .. 14 empty cells: the searches below them run deep (leaves with more than four changed cells).
    Colours follow (x + 2 y) mod 4 < 2 -- no line of that pattern is longer than two -- played in alternation."""
    rs, out = np.random.RandomState(seed), []
    while len(out) < count:
        empty = set(rs.permutation(B * B)[:rs.randint(5, 15)].tolist())
        cells = [[], []]
        for c in rs.permutation(B * B).tolist():
            if c not in empty:
                y, x = divmod(c, B)
                cells[0 if (x + 2 * y) % 4 < 2 else 1].append(c)
        e = RefGomoku(B, 5)
        for k in range(2 * min(len(cells[0]), len(cells[1]))):
            e.step(cells[k % 2][k // 2])
            assert not e.game_end_winner()[0]
        out.append(e)
    return out


total = 0
for B, G, sims in ((15, 700, 150), (11, 560, 100), (12, 300, 120), (13, 520, 100), (14, 300, 120), (16, 600, 100), (15, 64, 400),
                   (-15, 96, 300), (-11, 96, 300), (-13, 64, 300), (-16, 64, 300)):   # (negative: deep positions)
    t0 = time.time()
    deep, B = B < 0, abs(B)
    net = T._net(B, seed=100 + B)
    base = deep_positions(B, 32, seed=B) if deep else T._positions(B, 5, 160, seed=B * 7 + G)
    envs = [base[i % len(base)] for i in range(G)]
    sample = list(range(0, G, max(1, G // 24)))
    dumps, stats = {}, None
    for shipped in (True, False):
        ev = HipNetEvaluator(net, B, 'cuda:0', max_boards=G)
        ev.resident_search = shipped
        ev.delta_trunk = shipped
        eng = MCTSEngine(B, 5, n_games=G, n_playout=sims, device='cuda:0', add_noise=True, noise_seed=B)
        assert ev.resident_ok(eng) == shipped
        T._set_roots(eng, envs)
        eng.set_noise_keys()
        rec = []
        for move in range(3):
            eng.simulate(ev, sims, use_graph=False)
            visits = eng.root_visits()
            rec.append(visits.copy())
            rec.append([T._whole_tree(eng, g) for g in sample])
            playing = visits.sum(axis=1) > 0
            moves = np.where(playing, visits.argmax(axis=1), -2).astype(np.int32)
            eng.advance(moves)
            _, ended = eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
            eng.set_active((playing & (np.asarray(ended) == 0)).astype(np.uint8))
        eng.check()
        if shipped:
            stats = ev.hip.delta_stats()
        dumps[shipped] = rec
        eng.close()
        ev.hip.close()
    for a, b in zip(dumps[True], dumps[False]):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b, 'board %d: the routes differ' % B
    leaves = stats['delta'] + stats['no_base']
    total += leaves
    print('%2dx%-2d %4d games x %3d simulations x 3 moves: equal (visits of every game, %d whole trees per move); %d leaves, %d (%.2f %%) without a base, '
          '%.2f conv3 / %.2f conv2 tiles and %.2f changed cells per leaf; %.1f s' % (
              B, B, G, sims, len(sample), leaves, stats['no_base'], 100.0 * stats['no_base'] / max(1, leaves), stats['tiles3'] / max(1, leaves),
              stats['tiles2'] / max(1, leaves), stats['cells'] / max(1, leaves), time.time() - t0), flush=True)

# ---- small boards: the resident search on the compact LDS grid (two games per CU, rounds) against the two-launch step on the 18 x 18 grid
for B, n_row, G, sims in ((3, 3, 700, 25), (5, 4, 600, 60), (6, 4, 600, 100), (7, 5, 560, 100), (6, 4, 130, 200)):
    t0 = time.time()
    net = T._net(B, seed=300 + B)
    base = T._positions(B, n_row, 120, seed=B * 11 + G)
    envs = [base[i % len(base)] for i in range(G)]
    sample = list(range(0, G, max(1, G // 24)))
    dumps = {}
    for shipped in (True, False):
        ev = HipNetEvaluator(net, B, 'cuda:0', max_boards=G)
        ev.resident_search = shipped
        eng = MCTSEngine(B, n_row, n_games=G, n_playout=sims, device='cuda:0', add_noise=True, noise_seed=B)
        assert ev.resident_ok(eng) == shipped and ev.resident_per_cu(eng) == 2
        T._set_roots(eng, envs)
        eng.set_noise_keys()
        rec = []
        for move in range(4):
            eng.simulate(ev, sims, use_graph=False)
            visits = eng.root_visits()
            rec.append(visits.copy())
            rec.append([T._whole_tree(eng, g) for g in sample])
            playing = visits.sum(axis=1) > 0
            moves = np.where(playing, visits.argmax(axis=1), -2).astype(np.int32)
            eng.advance(moves)
            _, ended = eng.step(np.where(moves >= 0, moves, -1).astype(np.int32))
            eng.set_active((playing & (np.asarray(ended) == 0)).astype(np.uint8))
        eng.check()
        dumps[shipped] = rec
        eng.close()
        ev.hip.close()
    for a, b in zip(dumps[True], dumps[False]):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b, 'board %d: the routes differ' % B
    total += G * sims * 4
    print('%dx%d (n = %d) %4d games x %3d simulations x 4 moves, compact-grid resident search against the two-launch step: equal (visits of every game, '
          '%d whole trees per move); %.1f s' % (B, B, n_row, G, sims, len(sample), time.time() - t0), flush=True)
print('soak ok: %d leaves' % total)
