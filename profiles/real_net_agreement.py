"""How often does the device's search, fed by ITS OWN network values, reproduce the visit counts the reference produced with torch-CPU
values?  (north_star's "bit-exact move selection" when the real net drives the search.)

tests/golden/g8_realnet.json.gz holds whole self-play games of the reference (AlphaZeroPlayer + AlphaZeroAgent.policy_value_fn on the
CPU, numpy_weights, injected uniforms: tests/golden/gen_golden.py): per ply the root's visit counts, the move and the smallest gap
between the two best finite UCT scores any selection of that search met (node.py:41-42, 75-88).  Here the same games are played on the
GPU through the reference's API (GameControl.start_self_play + AlphaZeroPlayer with the hand-written evaluator) and compared ply by
ply until the first ply whose visit vector differs: from there on the two games are different games (another move, or another reused
subtree).  The network outputs agree to ~1e-7 (tests/test_gpu_parity.py: 1e-4 bound), the tree arithmetic is bit-exact given the
values, so a divergence can only come from a selection whose two best scores lay closer than the value difference; the report lists,
per divergence, the smallest gap the reference recorded up to that ply.

    python profiles/real_net_agreement.py > profiles/r06/real_net_agreement.txt      (on an MI355X)
"""
import gzip
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class Injected(object):
    """numpy.random.choice(acts, p=probs) from a recorded uniform (numpy's legacy algorithm: inverse CDF, side='right')."""

    def __init__(self, us):
        self.us = list(us)
        self.real = np.random.choice

    def __call__(self, acts, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return np.asarray(acts)[cdf.searchsorted(self.us.pop(0), side='right')]


def load_games():
    with gzip.open(os.path.join(REPO, 'tests', 'golden', 'g8_realnet.json.gz'), 'rb') as f:
        return json.loads(f.read().decode())['games']


def play_on_device(game, device='cuda:0'):
    """-> (plies that agree before the first difference, plies compared, record of the first difference or None)"""
    import torch
    from oracle.evaluators import numpy_weights   # (the weights' definition: test infrastructure, like this script)
    from rlzero_amd.games import GameControl, GomokuEnv
    from rlzero_amd.games.gomoku.alphazero_agent import AlphaZeroAgent
    from rlzero_amd.mcts import AlphaZeroPlayer
    B, n = game['B'], game['n']
    agent = AlphaZeroAgent(B, device=device)
    agent.policy_value_net.load_state_dict({k: torch.from_numpy(v) for k, v in numpy_weights(B, game['weights_seed']).items()})
    plies = game['plies']
    inj = Injected([float.fromhex(p['u']) for p in plies])
    seen = []

    class Stop(Exception):
        pass

    np.random.choice = inj
    player = AlphaZeroPlayer(agent.policy_value_fn, n_playout=game['n_playout'], c_puct=game['c_puct'], is_selfplay=True)
    try:
        real = player.mcts.simulate

        def spy(env, temperature=1e-3):
            acts, probs = real(env, temperature)
            visits = player.mcts._engine.root_visits()[0]
            k = len(seen)
            mine = [int(visits[a]) for a in acts]
            seen.append(mine)
            if k >= len(plies) or list(acts) != plies[k]['acts'] or mine != plies[k]['N']:
                raise Stop()
            return acts, probs

        player.mcts.simulate = spy
        env = GomokuEnv(B, n)
        try:
            GameControl(env).start_self_play(player, temperature=game['T'])
        except Stop:
            pass
    finally:
        np.random.choice = inj.real
        eng = getattr(player.mcts, '_engine', None)
        if eng is not None:
            eng.close()
    agree = len(seen) if (len(seen) == len(plies) and seen[-1] == plies[-1]['N']) else len(seen) - 1
    if agree == len(plies):
        return agree, len(plies), None
    k = agree
    gaps = [float.fromhex(p['min_gap']) for p in plies[:k + 1] if p['min_gap'] is not None]
    ref_n = plies[k]['N'] if k < len(plies) else None
    moved = sum(abs(a - b) for a, b in zip(seen[k], ref_n)) // 2 if ref_n is not None and len(ref_n) == len(seen[k]) else None
    return agree, len(plies), {'ply': k, 'smallest_gap_so_far': min(gaps) if gaps else None, 'gap_of_the_ply': plies[k]['min_gap'] and float.fromhex(plies[k]['min_gap']),
                               'visits_moved': moved}


def report(out=sys.stdout):
    games = load_games()
    total_agree = total = whole = 0
    rows = []
    for i, g in enumerate(games):
        agree, n, first = play_on_device(g)
        total_agree += agree
        total += n
        whole += first is None
        rows.append((i, g, agree, n, first))
        out.write('game %2d  %dx%d  %3d playouts  %2d plies: %s\n' % (
            i, g['B'], g['B'], g['n_playout'], n,
            'every ply identical (visit vectors, moves, winner)' if first is None else
            'identical up to ply %d; at ply %d %s visits sit on other children; smallest gap between the two best UCT scores the reference met: '
            'in that search %.3g, up to that ply %.3g' % (first['ply'], first['ply'], first['visits_moved'], first['gap_of_the_ply'] if first['gap_of_the_ply'] is not None else float('nan'),
                                                          first['smallest_gap_so_far'] if first['smallest_gap_so_far'] is not None else float('nan'))))
        out.flush()
    out.write('\n%d of %d games identical from the first ply to the last; %d of %d plies identical before a game\'s first difference '
              '(%.1f %%)\n' % (whole, len(games), total_agree, total, 100.0 * total_agree / max(1, total)))
    return rows


if __name__ == '__main__':
    report()
