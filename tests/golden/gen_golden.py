#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only).  The reference can
not travel to the GPU box, so its outputs are committed here as data:

  g1_rules.json.gz  rules: per-ply (action, ended, winner, legal count, 4 obs planes as
                   bit masks) of random play-outs + hand-made edge positions
  g2_search.json.gz search: root / child (N, W-as-hex-fp64, pi) and full tree dumps for the
                   synthetic evaluators v0 / vlin, tree reuse
  g3_games.json.gz  whole self-play games and two-player games with injected uniforms
  g2_netleaf.json.gz a search driven by the real net on CPU: per-simulation leaf values
  g4_net.npz       PolicyValueNet outputs for deterministic numpy weights
  g5_equi.npz      TrainPipeline.get_equi_data for one asymmetric sample
  g7_train.json.gz  the reference's TrainPipeline.run() (tools/train_alphazero.py) for 4 batches at 50 playouts: per batch
                   episode_len, buffer length, the moves of the game, and every number policy_update prints
                   (kl, lr_multiplier, loss, entropy, explained variances); search driven by vlin + injected uniforms
                   (exactly reproducible data), learner = the reference's AlphaZeroAgent.learn on numpy_weights
  g8_realnet.json.gz whole self-play games of the reference's AlphaZeroPlayer with ITS OWN evaluator -- AlphaZeroAgent.policy_value_fn,
                   torch on the CPU, numpy_weights -- at 6x6 / 400 playouts (the script's default, tools/train_alphazero.py:21-31) and
                   9x9 / 200: per ply the root's visit counts, the move, the uniform, and the SMALLEST GAP between the two best
                   finite UCT scores any selection of that search met (node.py:41-42, 75-88: a device whose leaf values differ
                   from torch's by 1e-7 may legitimately take the other child only where that gap is tiny)
  g6_rollout.json.gz pure-MCTS opponent (RolloutMCTS / RolloutPlayer) with np.random.rand drawn from a
                   recorded private stream: root statistics, chosen moves, a full duel

Usage:  python tests/golden/gen_golden.py        (rewrites the files next to it)

Only reference *outputs* are stored; no reference source text is copied.
"""
import gzip
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.path.insert(1, REPO)

_g = types.ModuleType('gymnasium')
_g.Env = type('Env', (), {})
sys.modules['gymnasium'] = _g

import torch  # noqa: E402

torch.set_num_threads(1)

from rlzero.games.gomoku import GameControl, GomokuEnv  # noqa: E402
from rlzero.games.gomoku.alphazero_agent import AlphaZeroAgent  # noqa: E402
from rlzero.mcts.alphazero_mcts import AlphaZeroMCTS, AlphaZeroPlayer  # noqa: E402
from rlzero.mcts.rollout_mcts import RolloutMCTS, RolloutPlayer  # noqa: E402

from oracle.evaluators import numpy_weights, v0, vlin  # noqa: E402  (pure functions)

EVALS = {'v0': v0, 'vlin': vlin}


def hexf(x):
    return float(x).hex()


def plane_bits(obs):
    """4 planes -> 4 python ints, bit (h*B+w) set where plane == 1.0."""
    out = []
    for p in range(4):
        flat = obs[p].reshape(-1)
        assert set(np.unique(flat)) <= {0.0, 1.0}
        out.append(hex(sum(1 << i for i, v in enumerate(flat) if v == 1.0)))
    return out


def new_env(B, n, moves=()):
    env = GomokuEnv(board_size=B, n_in_row=n)
    env.reset()
    for m in moves:
        env.step(int(m))
    return env


# ---------------------------------------------------------------------------- G1
def trace_moves(B, n, moves):
    env = new_env(B, n)
    plies = []
    for m in moves:
        env.step(int(m))
        ended, winner = env.game_end_winner()
        won, who = env.has_a_winner()
        plies.append({
            'a': int(m), 'ended': bool(ended), 'winner': int(winner), 'won': bool(won),
            'who': int(who), 'n_legal': len(env.leagel_actions()),
            'to_move': int(env.current_player()), 'last': int(env.last_move),
            'obs': plane_bits(env.current_state()),
        })
        if ended:
            break
    return plies


def gen_g1():
    rs = np.random.RandomState(20250217)
    out = {'random': [], 'handmade': []}
    for B, n, reps in ((3, 3, 12), (6, 4, 8), (9, 5, 6), (15, 5, 4), (8, 5, 3)):
        for _ in range(reps):
            perm = rs.permutation(B * B)
            plies = trace_moves(B, n, perm)
            out['random'].append({'B': B, 'n': n, 'plies': plies})

    def line_case(B, n, line, name, extra_p0=(), length=None):
        """player 0 plays ``line`` (last cell last), player 1 plays harmless cells."""
        line = list(line)
        for attempt in range(200):
            free = [c for c in range(B * B) if c not in line and c not in extra_p0]
            p1 = list(rs.choice(free, size=len(line) + len(extra_p0) - 1, replace=False))
            p0 = list(extra_p0) + line
            seq = []
            for i, c in enumerate(p0):
                seq.append(c)
                if i < len(p1):
                    seq.append(p1[i])
            plies = trace_moves(B, n, seq)
            # premature end (p1 won by accident or p0 earlier than the last stone)
            if len(plies) == len(seq):
                out['handmade'].append({'B': B, 'n': n, 'name': name, 'plies': plies})
                return
        raise RuntimeError('could not build ' + name)

    for B, n in ((3, 3), (6, 4), (9, 5), (15, 5)):
        last = B - n
        rows = sorted({0, last // 2, B - 1})
        for h in rows:  # horizontal at left edge, middle, right edge
            for w in sorted({0, last // 2, last}):
                line_case(B, n, [h * B + w + j for j in range(n)], 'row h%d w%d' % (h, w))
        for w in rows:  # vertical
            for h in sorted({0, last}):
                line_case(B, n, [(h + j) * B + w for j in range(n)], 'col h%d w%d' % (h, w))
        for h, w in ((0, 0), (last, last), (0, last), (last, 0)):  # diagonal
            line_case(B, n, [(h + j) * B + w + j for j in range(n)], 'diag h%d w%d' % (h, w))
        for h, w in ((0, B - 1), (last, n - 1), (0, n - 1), (last, B - 1)):  # anti-diagonal
            line_case(B, n, [(h + j) * B + w - j for j in range(n)], 'anti h%d w%d' % (h, w))
        # win completed by a stone in the MIDDLE of the line (last stone not at an end)
        mid = [1 * B + j for j in range(n)] if B > 3 else [3, 5, 4]
        if B > 3:
            mid = mid[:n // 2] + mid[n // 2 + 1:] + [mid[n // 2]]
        line_case(B, n, mid, 'row completed in the middle')
    # overline: six in a row at n=5 (counts as a win: any 5 consecutive cells)
    six = [2 * 9 + w for w in (1, 2, 3, 5, 6, 4)]
    line_case(9, 5, six, 'overline 6 completed in the middle')
    six = [7 * 15 + w for w in (9, 10, 11, 13, 14, 12)]
    line_case(15, 5, six, 'overline 6 at the right edge')
    # row wrap: cells B-2 .. B+2 are contiguous indices but NOT a line
    for B, n in ((9, 5), (15, 5), (6, 4)):
        wrap = [B - (n // 2) + j for j in range(n)]
        line_case(B, n, wrap, 'row-wrap is not a win')
        # anti-diagonal wrap: stride B-1 starting at column < n-1
        wrap = [1 + j * (B - 1) for j in range(n)]
        line_case(B, n, wrap, 'anti-diagonal wrap is not a win')
        wrap = [(B - 2) + j * (B + 1) for j in range(n)]
        line_case(B, n, wrap, 'diagonal wrap is not a win')
    # full-board tie on 3x3
    out['handmade'].append({'B': 3, 'n': 3, 'name': 'full-board tie',
                            'plies': trace_moves(3, 3, [0, 1, 2, 4, 3, 5, 7, 6, 8])})
    return out


# ---------------------------------------------------------------------------- G2
def ref_tree_dump(root):
    out = []
    stack = [((), root)]
    while stack:
        path, node = stack.pop()
        if node.explore_count > 0 or not path:
            out.append([list(path), int(node.explore_count), hexf(node.total_reward)])
        for a, kid in node._children.items():
            if kid.explore_count > 0:
                stack.append((path + (int(a), ), kid))
    out.sort(key=lambda e: e[0])
    return out


def dump_digest(dump):
    h = hashlib.sha1()
    for path, n, w in dump:
        h.update(repr((tuple(path), n, w)).encode())
    return h.hexdigest()


def root_record(mcts, acts, probs):
    root = mcts._root
    kids = list(root._children.items())
    assert tuple(a for a, _ in kids) == tuple(acts)
    return {
        'root_N': int(root.explore_count), 'root_W': hexf(root.total_reward),
        'acts': [int(a) for a in acts],
        'N': [int(k.explore_count) for _, k in kids],
        'W': [hexf(k.total_reward) for _, k in kids],
        'pi': [hexf(p) for p in probs],
    }


def gen_g2():
    cases = []
    spec = [
        # B, n, pre-moves, evaluator, n_playout, temperature, full dump?
        (3, 3, [], 'v0', 25, 1.0, True),
        (3, 3, [4, 0, 2], 'v0', 200, 1.0, True),
        (3, 3, [], 'vlin', 25, 1.0, True),
        (3, 3, [4, 0, 2], 'vlin', 200, 1.0, True),
        (3, 3, [4, 0, 2, 6, 3], 'vlin', 300, 1.0, True),   # deep, many terminal leaves
        (3, 3, [0, 4, 8, 2, 6, 3, 5], 'vlin', 64, 1e-3, True),  # 2 empty cells
        (3, 3, [0, 1, 2, 4, 3, 5, 7, 6], 'vlin', 10, 1.0, True),  # 1 empty cell -> tie leaf
        (6, 4, [], 'vlin', 400, 1.0, True),
        (6, 4, [14, 15, 20, 21, 8], 'vlin', 400, 1e-3, True),
        (6, 4, [14, 0, 15, 1, 16], 'vlin', 600, 1.0, True),  # immediate threats
        (9, 5, [], 'vlin', 200, 1.0, True),
        (9, 5, [40, 41, 31, 49, 22, 58], 'vlin', 500, 1.0, True),
        (8, 5, [27, 28], 'v0', 150, 1.0, True),
        (15, 5, [], 'vlin', 800, 1.0, False),
        (15, 5, [112, 113, 97, 127, 98, 96, 128], 'vlin', 800, 1.0, False),
        (15, 5, [112], 'v0', 300, 1.0, False),
    ]
    for B, n, pre, ev, sims, T, full in spec:
        env = new_env(B, n, pre)
        mcts = AlphaZeroMCTS(EVALS[ev], n_playout=sims, c_puct=5)
        acts, probs = mcts.simulate(env, temperature=T)
        rec = {'B': B, 'n': n, 'pre': pre, 'eval': ev, 'n_playout': sims, 'c_puct': 5,
               'T': T}
        rec.update(root_record(mcts, acts, probs))
        dump = ref_tree_dump(mcts._root)
        rec['n_nodes'] = len(dump)
        rec['tree_sha1'] = dump_digest(dump)
        if full:
            rec['tree'] = dump
        cases.append(rec)
    # other c_puct values (0 -> pure exploitation, ties everywhere)
    for c in (0.0, 0.5, 1.25):
        env = new_env(6, 4, [14, 15])
        mcts = AlphaZeroMCTS(vlin, n_playout=300, c_puct=c)
        acts, probs = mcts.simulate(env, temperature=1.0)
        rec = {'B': 6, 'n': 4, 'pre': [14, 15], 'eval': 'vlin', 'n_playout': 300,
               'c_puct': c, 'T': 1.0}
        rec.update(root_record(mcts, acts, probs))
        dump = ref_tree_dump(mcts._root)
        rec['n_nodes'] = len(dump)
        rec['tree_sha1'] = dump_digest(dump)
        rec['tree'] = dump
        cases.append(rec)
    return {'cases': cases}


# ---------------------------------------------------------------------------- G3
class InjectedChoice(object):
    """Stand-in for numpy.random.choice(acts, p=probs) that draws its uniform from a
    private stream and records it (numpy's legacy algorithm: inverse CDF, side='right')."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.used = []
        self.real = np.random.choice

    def __call__(self, acts, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        while True:
            u = float(self.rs.random_sample())
            idx = int(cdf.searchsorted(u, side='right'))
            edges = np.concatenate(([0.0], cdf))
            if np.min(np.abs(edges - u)) > 1e-9:  # stay away from cdf edges
                break
        self.used.append(u)
        return np.asarray(acts)[idx]


def check_choice_equivalence():
    rs_a = np.random.RandomState(7)
    rs_b = np.random.RandomState(7)
    for _ in range(1000):
        k = rs_a.randint(1, 40)
        rs_b.randint(1, 40)
        p = rs_a.dirichlet(np.ones(k))
        rs_b.dirichlet(np.ones(k))
        acts = tuple(range(3, 3 + k))
        want = rs_a.choice(acts, p=p)
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        got = np.asarray(acts)[cdf.searchsorted(rs_b.random_sample(), side='right')]
        assert want == got


def selfplay_record(B, n, ev, sims, T, seed, max_plies=None):
    inj = InjectedChoice(seed)
    np.random.choice = inj
    try:
        env = GomokuEnv(board_size=B, n_in_row=n)
        game = GameControl(env)
        player = AlphaZeroPlayer(EVALS[ev], n_playout=sims, c_puct=5, is_selfplay=True)
        # instrument: record the root children after every simulate()
        per_ply = []
        real_sim = player.mcts.simulate

        def spy(game_env, temperature=1e-3):
            acts, probs = real_sim(game_env, temperature)
            per_ply.append(root_record(player.mcts, acts, probs))
            return acts, probs

        player.mcts.simulate = spy
        winner, data = game.start_self_play(player, temperature=T)
        data = list(data)
    finally:
        np.random.choice = inj.real
    plies = []
    for i, (state, pi, z) in enumerate(data):
        rec = per_ply[i]
        rec['u'] = hexf(inj.used[i])
        rec['obs'] = plane_bits(state)
        rec['z'] = float(z)
        rec['pi_full'] = [hexf(x) for x in pi]
        plies.append(rec)
    # recover the moves from the final env
    moves = [int(m) for m in env.states.keys()]
    assert len(moves) == len(plies)
    root = player.mcts._root
    return {'B': B, 'n': n, 'eval': ev, 'n_playout': sims, 'T': T, 'winner': int(winner),
            'moves': moves, 'plies': plies,
            'root_after_reset': [int(root.explore_count), len(root._children)]}


def twoplayer_record(B, n, sims, seed):
    """start_play between two non-self-play AlphaZeroPlayers (two draws per move, the
    second is used; root reset every move: alphazero_mcts.py:153-158)."""
    inj = InjectedChoice(seed)
    np.random.choice = inj
    try:
        env = GomokuEnv(board_size=B, n_in_row=n)
        game = GameControl(env)
        p1 = AlphaZeroPlayer(vlin, n_playout=sims, c_puct=5)
        p2 = AlphaZeroPlayer(v0, n_playout=sims // 2, c_puct=5)
        winner = game.start_play(p1, p2, start_player=0, is_shown=0)
    finally:
        np.random.choice = inj.real
    return {'B': B, 'n': n, 'n_playout': [sims, sims // 2], 'evals': ['vlin', 'v0'],
            'winner': int(winner), 'moves': [int(m) for m in env.states.keys()],
            'u': [hexf(u) for u in inj.used]}


def gen_g3():
    check_choice_equivalence()
    games = [
        selfplay_record(3, 3, 'vlin', 25, 1.0, 1),
        selfplay_record(3, 3, 'v0', 25, 1.0, 2),
        selfplay_record(3, 3, 'vlin', 60, 1e-3, 3),
        selfplay_record(6, 4, 'vlin', 100, 1.0, 4),
        selfplay_record(6, 4, 'vlin', 400, 1.0, 5),
        selfplay_record(9, 5, 'vlin', 60, 1.0, 6),
        selfplay_record(8, 5, 'v0', 40, 1.0, 7),
    ]
    duels = [twoplayer_record(3, 3, 30, 11), twoplayer_record(6, 4, 80, 12)]
    return {'selfplay': games, 'duels': duels}


# ---------------------------------------------------------------------------- G8
def realnet_game(B, n, sims, seed, wseed):
    """One self-play game of the reference's player with its own torch-CPU evaluator; TreeNode.select is watched (not changed)."""
    from rlzero.mcts.node import TreeNode
    agent = AlphaZeroAgent(B)
    load_numpy_weights(agent, B, wseed)
    inj = InjectedChoice(seed)
    np.random.choice = inj
    real_select = TreeNode.select
    gap = [float('inf'), 0]   # of the search in progress: smallest gap, selections among finite scores

    def watching_select(node, c_puct):
        best, second = float('-inf'), float('-inf')
        for child in node._children.values():
            s_ = child.uct_value(c_puct)
            if s_ > best:
                best, second = s_, best
            elif s_ > second:
                second = s_
        if best != float('inf') and second != float('-inf'):
            gap[0] = min(gap[0], best - second)
            gap[1] += 1
        return real_select(node, c_puct)

    TreeNode.select = watching_select
    try:
        env = GomokuEnv(board_size=B, n_in_row=n)
        game = GameControl(env)
        player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=sims, c_puct=5, is_selfplay=True)
        per_ply = []
        real_sim = player.mcts.simulate

        def spy(game_env, temperature=1e-3):
            gap[0], gap[1] = float('inf'), 0
            with torch.no_grad():
                acts, probs = real_sim(game_env, temperature)
            root = player.mcts._root
            per_ply.append({'acts': [int(a) for a in acts], 'N': [int(root._children[a].explore_count) for a in acts],
                            'root_N': int(root.explore_count), 'min_gap': hexf(gap[0]) if gap[0] != float('inf') else None,
                            'value_selections': gap[1]})
            return acts, probs

        player.mcts.simulate = spy
        winner, data = game.start_self_play(player, temperature=1.0)
        data = list(data)
    finally:
        np.random.choice = inj.real
        TreeNode.select = real_select
    moves = [int(m) for m in env.states.keys()]
    assert len(moves) == len(per_ply) == len(inj.used)
    for rec, u, mv in zip(per_ply, inj.used, moves):
        rec['u'] = hexf(u)
        rec['move'] = mv
    return {'B': B, 'n': n, 'n_playout': sims, 'c_puct': 5, 'T': 1.0, 'weights_seed': wseed, 'winner': int(winner), 'moves': moves, 'plies': per_ply}


def gen_g8():
    np.random.seed(8)   # (the reference's Dirichlet noise draws from the global stream: it never reaches the selection)
    games = []
    for B, n, sims, count, wseed in ((6, 4, 400, 8, 301), (9, 5, 200, 8, 302)):
        for k in range(count):
            games.append(realnet_game(B, n, sims, 8000 + 100 * B + k, wseed))
    return {'games': games}


# ---------------------------------------------------------------------------- net
def load_numpy_weights(agent, B, seed):
    w = numpy_weights(B, seed)
    sd = agent.policy_value_net.state_dict()
    assert list(sd.keys()) == list(w.keys()), (list(sd.keys()), list(w.keys()))
    agent.policy_value_net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})


def gen_netleaf():
    """Search on 6x6 driven by the real net (CPU): per-simulation (path, leaf value)."""
    out = []
    for B, n, pre, sims, seed in ((6, 4, [], 150, 101), (6, 4, [14, 15, 20], 120, 102),
                                  (3, 3, [4], 80, 103), (9, 5, [40], 120, 104)):
        agent = AlphaZeroAgent(B)
        load_numpy_weights(agent, B, seed)
        env = new_env(B, n, pre)
        mcts = AlphaZeroMCTS(agent.policy_value_fn, n_playout=sims, c_puct=5)
        log = []
        real_pvf = mcts.policy_value_fn

        def spy(e, _log=log, _pvf=real_pvf):
            probs, v = _pvf(e)
            _log.append([[int(m) for m in e.states.keys()], hexf(v)])
            return probs, v

        mcts.policy_value_fn = spy
        with torch.no_grad():
            acts, probs = mcts.simulate(env, temperature=1.0)
        rec = {'B': B, 'n': n, 'pre': pre, 'n_playout': sims, 'seed': seed, 'c_puct': 5,
               'T': 1.0, 'leaves': log}
        rec.update(root_record(mcts, acts, probs))
        dump = ref_tree_dump(mcts._root)
        rec['tree'] = dump
        out.append(rec)
    return {'cases': out}


def gen_g4():
    arrays = {}
    rs = np.random.RandomState(4)
    for B, n, seed in ((3, 3, 31), (6, 4, 32), (9, 5, 33), (15, 5, 34)):
        agent = AlphaZeroAgent(B)
        load_numpy_weights(agent, B, seed)
        obs = []
        for i in range(16):
            env = new_env(B, n)
            n_moves = 0 if i == 0 else rs.randint(1, B * B)
            for m in rs.permutation(B * B)[:n_moves]:
                env.step(int(m))
                if env.game_end_winner()[0]:
                    break
            obs.append(env.current_state())
        obs = np.array(obs)
        with torch.no_grad():
            logp, v = agent.policy_value_net(torch.from_numpy(obs).float())
        arrays['B%d_seed' % B] = np.array(seed)
        arrays['B%d_obs' % B] = obs.astype(np.uint8)
        arrays['B%d_logp' % B] = logp.numpy()
        arrays['B%d_value' % B] = v.numpy()
        # evaluator plug-in contract (alphazero_agent.py:31-46) on the first obs
        env = new_env(B, n, [0] if B * B > 1 else [])
        probs, value = agent.policy_value_fn(env)
        probs = list(probs)
        arrays['B%d_pvf_acts' % B] = np.array([a for a, _ in probs])
        arrays['B%d_pvf_probs' % B] = np.array([p for _, p in probs], dtype=np.float32)
        arrays['B%d_pvf_value' % B] = np.array(value, dtype=np.float64)
    return arrays


def gen_g5():
    spec = importlib.util.spec_from_file_location('ref_train',
                                                  '/root/reference/tools/train_alphazero.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fake = types.SimpleNamespace(board_size=4)
    rs = np.random.RandomState(5)
    state = rs.randint(0, 2, size=(4, 4, 4)).astype(np.float64)
    pi = rs.dirichlet(np.ones(16))
    out = mod.TrainPipeline.get_equi_data(fake, [(state, pi, 1.0)])
    return {'state': state, 'pi': pi,
            'equi_states': np.array([s for s, _, _ in out]),
            'equi_pis': np.array([p for _, p, _ in out]),
            'equi_z': np.array([z for _, _, z in out])}


def gen_g6():
    """RolloutMCTS with numpy.random.rand redirected to RandomState(seed).rand (so the oracle can
    replay the identical stream): root children after simulate, the move, full tree dumps."""
    real_rand = np.random.rand
    cases = []
    try:
        for B, n, pre, sims, seed in ((3, 3, [], 60, 61), (3, 3, [4, 0, 2], 200, 62), (6, 4, [], 150, 63),
                                      (6, 4, [14, 15, 20, 21, 8], 200, 64), (9, 5, [40, 41], 100, 65),
                                      (3, 3, [0, 1, 2, 4, 3, 5, 7], 40, 66)):
            rs = np.random.RandomState(seed)
            np.random.rand = rs.rand
            env = new_env(B, n, pre)
            mcts = RolloutMCTS(n_playout=sims, c_puct=5)
            move = mcts.simulate(env)
            root = mcts._root
            kids = list(root._children.items())
            dump = ref_tree_dump(root)
            cases.append({'B': B, 'n': n, 'pre': pre, 'n_playout': sims, 'seed': seed, 'move': int(move),
                          'root_N': int(root.explore_count), 'root_W': hexf(root.total_reward),
                          'acts': [int(a) for a, _ in kids], 'N': [int(k.explore_count) for _, k in kids],
                          'W': [hexf(k.total_reward) for _, k in kids], 'tree': dump,
                          'n_rand_left': hexf(rs.rand())})
        # a whole game RolloutPlayer vs RolloutPlayer through GameControl.start_play
        rs = np.random.RandomState(77)
        np.random.rand = rs.rand
        env = GomokuEnv(board_size=3, n_in_row=3)
        winner = GameControl(env).start_play(RolloutPlayer(n_playout=40), RolloutPlayer(n_playout=25),
                                             start_player=0, is_shown=0)
        duel = {'B': 3, 'n': 3, 'n_playout': [40, 25], 'seed': 77, 'winner': int(winner),
                'moves': [int(m) for m in env.states.keys()]}
    finally:
        np.random.rand = real_rand
    return {'cases': cases, 'duel': duel}


def gen_g7():
    """TrainPipeline.run() of the reference, 4 batches.  The self-play search is driven by the exactly representable
    vlin evaluator with injected move uniforms, so the collected (state, pi, z) data are reproducible bit for bit on
    any host; the learner is the reference's own (AlphaZeroAgent.learn on CPU, deterministic numpy weights), so
    loss / entropy / kl pin policy_update + learn (train_alphazero.py:92-137, alphazero_agent.py:59-86)."""
    import contextlib
    import io
    import random
    import tempfile
    spec = importlib.util.spec_from_file_location('ref_train7', '/root/reference/tools/train_alphazero.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seed, wseed, n_playout, batches = 70, 71, 50, 4
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    cwd = os.getcwd()
    inj = InjectedChoice(seed)
    records = []
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        np.random.choice = inj
        try:
            pipe = mod.TrainPipeline()
            pipe.n_playout = n_playout
            pipe.game_batch_num = batches
            load_numpy_weights(pipe.alphazero_agent, pipe.board_size, wseed)
            pipe.mcts_player = AlphaZeroPlayer(vlin, n_playout=n_playout, c_puct=pipe.c_puct, is_selfplay=True)
            real_collect, real_update = pipe.collect_selfplay_data, pipe.policy_update

            def collect(n_games=1):
                real_collect(n_games)
                records.append({'episode_len': int(pipe.episode_len), 'buffer_len': len(pipe.data_buffer),
                                'moves': [int(m) for m in pipe.board.states.keys()], 'u_used': len(inj.used)})

            def update():
                loss, entropy = real_update()
                records[-1].update({'loss': hexf(loss), 'entropy': hexf(entropy),
                                    'lr_multiplier': hexf(pipe.lr_multiplier)})
                return loss, entropy

            pipe.collect_selfplay_data, pipe.policy_update = collect, update
            out = io.StringIO()
            with contextlib.redirect_stdout(out):
                pipe.run()
        finally:
            np.random.choice = inj.real
            os.chdir(cwd)
    sd = pipe.alphazero_agent.policy_value_net.state_dict()
    return {'seed': seed, 'weight_seed': wseed, 'B': pipe.board_size, 'n': pipe.n_in_row, 'n_playout': n_playout,
            'c_puct': pipe.c_puct, 'temperature': pipe.temperature, 'batch_size': pipe.batch_size, 'epochs': pipe.epochs,
            'buffer_size': pipe.buffer_size, 'batches': records, 'u': [hexf(u) for u in inj.used],
            'stdout': out.getvalue().splitlines(),
            'final_weights_abs_sum': {k: hexf(float(v.double().abs().sum())) for k, v in sd.items()},
            'final_conv1_head': [hexf(float(x)) for x in sd['conv1.weight'].flatten()[:8]]}


def write_json(name, obj):
    path = os.path.join(HERE, name + '.gz')
    with gzip.GzipFile(path, 'wb', mtime=0) as f:  # mtime=0: reproducible bytes
        f.write(json.dumps(obj, separators=(',', ':')).encode())
    print('%-18s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def main():
    if sys.argv[1:] == ['g8']:   # (this fixture alone: the others do not depend on it)
        write_json('g8_realnet.json', gen_g8())
        return
    np.random.seed(0)
    write_json('g1_rules.json', gen_g1())
    write_json('g2_search.json', gen_g2())
    write_json('g3_games.json', gen_g3())
    write_json('g2_netleaf.json', gen_netleaf())
    write_json('g6_rollout.json', gen_g6())
    write_json('g7_train.json', gen_g7())
    write_json('g8_realnet.json', gen_g8())
    np.savez_compressed(os.path.join(HERE, 'g4_net.npz'), **gen_g4())
    np.savez_compressed(os.path.join(HERE, 'g5_equi.npz'), **gen_g5())
    for name in ('g4_net.npz', 'g5_equi.npz'):
        print('%-18s %8.1f KB' % (name, os.path.getsize(os.path.join(HERE, name)) / 1024.0))


if __name__ == '__main__':
    main()
