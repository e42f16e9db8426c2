"""INTEGRATION.md section 2 shows the bindings a maintainer of the reference would write against the C ABI alone (ctypes + torch for
device memory).  They are executed here as they stand in the document: the one-game search against the oracle's tree, the
device-driven self-play loop against the oracle's games."""
import os
import re

import numpy as np
import pytest
from conftest import REPO

from oracle import evaluators as ev
from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import RefPlayer, RefSearch, inverse_cdf_choice, self_play_game

pytestmark = pytest.mark.gpu


def _stub_namespace():
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    blocks = [b for b in re.findall(r'```python\n(.*?)```', text, flags=re.S) if 'class HipSearch' in b or 'def self_play' in b or 'def search_with_the_hand_written_net' in b]
    assert len(blocks) == 3 and 'class HipSearch' in blocks[0] and 'def search_with_the_hand_written_net' in blocks[1] and 'def self_play' in blocks[2]
    lib_path = os.path.join(REPO, 'rlzero_amd', 'librlzero_hip.so')
    ns = {}
    exec(blocks[0].replace("ctypes.CDLL('librlzero_hip.so')", 'ctypes.CDLL(%r)' % lib_path), ns)
    from rlzero_amd.selfplay import move_uniform
    ns['move_uniform'] = lambda seed, gid, ply: float(move_uniform(seed, gid, int(ply)))
    exec(blocks[1], ns)
    exec(blocks[2], ns)
    return ns


def test_the_one_game_stub_searches_like_the_oracle():
    import torch
    ns = _stub_namespace()
    B, n, sims = 6, 4, 150

    def net(obs):   # uniform policy, value 0: SURVEY.md Appendix B's v0 (the reference's rule never reads the priors)
        return torch.full((1, B * B), -float(np.log(B * B)), device=obs.device), torch.zeros(1, 1, device=obs.device)

    search = ns['HipSearch'](B, n, sims, 5.0, net)
    env = RefGomoku.from_moves(B, n, [14, 15, 20])
    acts, probs = search.simulate(env, 1.0)
    ref = RefSearch(ev.v0, sims, 5)
    want_acts, want_probs = ref.simulate(env, 1.0)
    assert tuple(acts) == tuple(want_acts) and np.max(np.abs(probs - want_probs)) <= 1e-12
    search.update_with_move(int(acts[int(np.argmax(probs))]))   # (update_with_move keeps the subtree: one more search from it)
    env.step(int(acts[int(np.argmax(probs))]))
    ref.update_with_move(int(want_acts[int(np.argmax(want_probs))]))
    a2, p2 = search.simulate(env, 1.0)
    w2, q2 = ref.simulate(env, 1.0)
    assert tuple(a2) == tuple(w2) and np.max(np.abs(p2 - q2)) <= 1e-12


def test_the_self_play_stub_plays_the_oracle_s_games():
    import ctypes
    import torch
    ns = _stub_namespace()
    lib, P, ok = ns['lib'], ns['P'], ns['ok']
    B, n, sims, slots, seed = 3, 3, 20, 4, 7
    cfg = ns['RzConfig'](lib.rz_abi_version(), 0, B, n, slots, sims, 0, 0, 5.0, 0.0, 0, 0, 0, 0, 0, 0)
    h = P()
    ok(lib.rz_create(ctypes.byref(cfg), ctypes.byref(h)))
    logp = torch.zeros(slots, B * B, device='cuda')
    value = torch.zeros(slots, device='cuda')
    st = P(torch.cuda.current_stream().cuda_stream)

    def net_eval(handle):   # one simulation step of every slot with the synthetic evaluator `vlin` (rz_eval_synthetic kind 1)
        ok(lib.rz_select_step(handle, None, st))
        ok(lib.rz_eval_synthetic(handle, 1, P(logp.data_ptr()), P(value.data_ptr()), st))
        ok(lib.rz_expand_backup(handle, P(logp.data_ptr()), P(value.data_ptr()), st))

    ids = list(range(10, 19))
    games = {gid: (moves, visits, winner) for gid, moves, visits, winner in ns['self_play'](h, net_eval, ids, slots, sims, B * B, seed=seed)}
    assert sorted(games) == ids
    from rlzero_amd.selfplay import move_uniform, visits_to_pi
    for gid, (moves, visits, winner) in games.items():
        us = move_uniform(seed, np.full(16, gid), np.arange(16))
        player = RefPlayer(ev.vlin, sims, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
        w, data, want_moves = self_play_game(RefGomoku(B, n), player, temperature=1.0)
        assert (w, want_moves) == (winner, moves)
        for (_, pi, _), counts in zip(data, visits):
            acts = np.nonzero(counts >= 0)[0]
            assert np.max(np.abs(pi[acts] - visits_to_pi(counts[acts], 1.0))) <= 1e-12
    lib.rz_destroy(h)


def test_the_resident_search_stub_with_and_without_receptive_fields():
    """INTEGRATION.md's third stub: one call searches every slot with the hand-written evaluator; with the roots' activations cached
    (rz_net_delta_*: leaves evaluated by their receptive fields, two games per CU) the visit counts are those of the full forward
    pass per leaf -- and those of this package's own engine wrapper on the same positions."""
    import ctypes
    import torch
    from rlzero_amd.engine import PARAM_ORDER
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    ns = _stub_namespace()
    lib, P, ok = ns['lib'], ns['P'], ns['ok']
    B, n, sims, slots = 15, 5, 120, 6
    torch.manual_seed(3)
    sd = PolicyValueNet(B).state_dict()
    arrays = [sd[name].detach().cpu().float().contiguous().numpy() for name in PARAM_ORDER]
    net = P()
    ok(lib.rz_net_create(B, B, B * B, 0, ctypes.byref(net)))
    ok(lib.rz_net_load(net, (ctypes.c_void_p * 16)(*[a.ctypes.data for a in arrays]), 16))
    rng = np.random.default_rng(5)
    stones = np.zeros((slots, 2, 4), dtype=np.uint64)
    to_move = np.zeros(slots, dtype=np.int32)
    last = np.full(slots, -1, dtype=np.int32)
    for g in range(slots):   # a few stones on every board (slot 0: the empty board)
        cells = rng.permutation(B * B)[:3 * g]
        for j, c in enumerate(cells):
            stones[g, j % 2, int(c) >> 6] |= np.uint64(1) << np.uint64(int(c) & 63)
        to_move[g], last[g] = len(cells) % 2, (int(cells[-1]) if len(cells) else -1)
    d_stones = torch.from_numpy(stones.view(np.int64)).cuda()
    d_tm, d_last = torch.from_numpy(to_move).cuda(), torch.from_numpy(last).cuda()
    st = P(torch.cuda.current_stream().cuda_stream)
    counts = {}
    for rf in (True, False):
        cfg = ns['RzConfig'](lib.rz_abi_version(), 0, B, n, slots, sims, 0, 0, 5.0, 0.0, 0, 0, 0, 0, 0, 0)
        h = P()
        ok(lib.rz_create(ctypes.byref(cfg), ctypes.byref(h)))
        ok(lib.rz_set_roots(h, P(d_stones.data_ptr()), P(d_tm.data_ptr()), P(d_last.data_ptr()), None, 1, st))
        counts[rf] = ns['search_with_the_hand_written_net'](h, net, slots, sims, B * B, receptive_field=rf).cpu().numpy()
        torch.cuda.synchronize()
        ok(lib.rz_destroy(h))
    assert np.array_equal(counts[True], counts[False]) and (counts[True].sum(axis=1) == sims - 1).all()
    ok(lib.rz_net_destroy(net))
