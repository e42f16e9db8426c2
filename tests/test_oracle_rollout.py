"""Pin oracle/rollout_ref.py (pure-MCTS opponent) against the reference's outputs: the
fixture was produced with numpy.random.rand redirected to RandomState(seed).rand, the oracle
replays the identical stream."""
import numpy as np

from oracle.gomoku_ref import RefGomoku
from oracle.mcts_ref import play_game, tree_dump
from oracle.rollout_ref import RefRolloutPlayer, RefRolloutSearch


def test_g6_rollout_search(g6):
    assert len(g6['cases']) >= 6
    for rec in g6['cases']:
        rs = np.random.RandomState(rec['seed'])
        env = RefGomoku.from_moves(rec['B'], rec['n'], rec['pre'])
        s = RefRolloutSearch(rec['n_playout'], 5, rand=rs.rand)
        move = s.simulate(env)
        assert move == rec['move']
        assert (s.root.n, float(s.root.w).hex()) == (rec['root_N'], rec['root_W'])
        assert list(s.root.acts) == rec['acts']
        assert [k.n for k in s.root.kids] == rec['N']
        assert [float(k.w).hex() for k in s.root.kids] == rec['W']
        got = {p: (n, float(w).hex()) for p, (n, w) in tree_dump(s.root).items()}
        assert got == {tuple(p): (n, w) for p, n, w in rec['tree']}
        # the oracle consumed exactly as many random numbers as the reference
        assert float(rs.rand()).hex() == rec['n_rand_left']


def test_g6_rollout_duel(g6):
    duel = g6['duel']
    rs = np.random.RandomState(duel['seed'])
    p1 = RefRolloutPlayer(duel['n_playout'][0], 5, rand=rs.rand)
    p2 = RefRolloutPlayer(duel['n_playout'][1], 5, rand=rs.rand)
    winner, moves = play_game(RefGomoku(duel['B'], duel['n']), p1, p2)
    assert (winner, moves) == (duel['winner'], duel['moves'])


def test_rollout_value_is_never_positive():
    """The reference's perspective rule (rollout_mcts.py:68-72): the winner has just moved, so a
    decisive rollout is worth -1 to the player to move afterwards; the backed-up value is +1."""
    rs = np.random.RandomState(0)
    s = RefRolloutSearch(50, 5, rand=rs.rand)
    values = [s.evaluate(RefGomoku.from_moves(6, 4, [14, 15])) for _ in range(50)]
    assert set(values) <= {-1.0, 0}
