#!/usr/bin/env python3
"""Round 6 soak, whole games: BatchedSelfPlay.run_device on the shipped layout (15x15, 800 simulations per move, 512 slots = one resident
lane of k_delta_res, the move step on the device, slots refilled) for 2048 games to their END, against ONE plain lane launched kernel by
kernel with the FULL-BOARD trunk on a sample of 96 of the same game ids (first-generation and refilled ones): moves, pi bits, winners.
And the same with 1536 slots (the launch in three rounds).    python3 profiles/soak_games_r06.py > profiles/r06/soak_games.txt"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet   # noqa: E402
from rlzero_amd.selfplay import BatchedSelfPlay   # noqa: E402

torch.manual_seed(3)
net = PolicyValueNet(15).to('cuda:0')
n_ids, sims = 2048, 800
sample = sorted(set(list(range(0, 512, 9)) + list(range(512, n_ids, 41)) + [511, n_ids - 1]))[:96]


def play(slots, ids, **kw):
    t0 = time.time()
    sp = BatchedSelfPlay.for_network(net, 15, 5, n_games=slots, n_playout=sims, seed=11, **kw)
    out = sp.run_device(ids) if not kw else sp.run(ids)
    for st in sp.check():
        assert st.reuse_dropped == 0 and st.max_slots_used < st.arena_slots
    lanes = len(sp.lanes)
    for lane in sp.lanes:
        lane.evaluator.hip.check_flags()
        lane.eng.close()
    return {t.game_id: t for t in out}, lanes, time.time() - t0


shipped, lanes, dt = play(512, range(n_ids))
assert sorted(shipped) == list(range(n_ids)) and lanes == 1
plies = [len(t.moves) for t in shipped.values()]
print('512 slots, %d games to the end: %.1f s (%.0f games / s with the warm-up), mean %.1f plies, longest %d, ties %d' % (
    n_ids, dt, n_ids / dt, np.mean(plies), max(plies), sum(t.winner == -1 for t in shipped.values())), flush=True)
rounds, lanes3, dt3 = play(1536, range(n_ids))
assert lanes3 == 1
for g in range(n_ids):
    a, b = shipped[g], rounds[g]
    assert a.moves == b.moves and a.winner == b.winner and np.array_equal(a.pis.view(np.uint64), b.pis.view(np.uint64)), g
print('1536 slots (three rounds per launch), the same %d games: every game equal (moves, pi bits, winner); %.1f s' % (n_ids, dt3), flush=True)
plain, _, dtp = play(len(sample), sample, lanes=1, use_graph=False, resident_search=False, delta_trunk=False)
for g in sample:
    a, b = shipped[g], plain[g]
    assert a.moves == b.moves and a.winner == b.winner and np.array_equal(a.pis.view(np.uint64), b.pis.view(np.uint64)), g
print('one plain lane, full-board trunk, kernel by kernel, %d of those games: equal (moves, pi bits, winner); %.1f s' % (len(sample), dtp), flush=True)
print('soak ok')
