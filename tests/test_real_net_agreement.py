"""The device's search with ITS OWN network values against the reference's own run with torch-CPU values (tests/golden/g8_realnet.json.gz:
whole self-play games at 6x6 / 400 and 9x9 / 200 playouts): every difference must sit behind a selection whose two best UCT scores
(node.py:41-42, 75-88) the reference itself recorded as a near-tie -- otherwise it is a bug, not rounding.  The figures themselves
(plies identical, first difference per game) are profiles/r06/real_net_agreement.txt, written by profiles/real_net_agreement.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles'))


def test_the_oracle_reproduces_the_reference_s_real_net_games_on_the_cpu():
    """Not a GPU test: the restatement (oracle/) with the torch-CPU forward on the same weights plays the reference's game ply for ply --
    visit vectors and moves -- for the shortest 6x6 game (the fixtures pin the oracle with the REAL evaluator too, not only with v0 / vlin)."""
    from real_net_agreement import load_games
    from oracle import evaluators as ev
    from oracle.gomoku_ref import RefGomoku
    from oracle.mcts_ref import RefPlayer, inverse_cdf_choice
    game = min((g for g in load_games() if g['B'] == 6), key=lambda g: len(g['plies']))
    pvf = ev.NetEvaluator(ev.numpy_weights(game['B'], game['weights_seed']), game['B'])
    env = RefGomoku(game['B'], game['n'])
    env.reset()
    player = RefPlayer(pvf, n_playout=game['n_playout'], c_puct=game['c_puct'], is_selfplay=True,
                       choice=inverse_cdf_choice(float.fromhex(p['u']) for p in game['plies']))
    for ply in game['plies']:
        acts, probs = player.mcts.simulate(env, game['T'])
        assert list(acts) == ply['acts'] and [kid.n for kid in player.mcts.root.kids] == ply['N']
        move = player.choice(acts, probs)
        assert int(move) == ply['move']
        player.mcts.update_with_move(move)
        env.step(move)
    assert env.game_end_winner() == (True, game['winner'])


@pytest.mark.gpu
def test_every_difference_from_the_reference_s_real_net_games_sits_behind_a_near_tie():
    from real_net_agreement import load_games, play_on_device
    games = load_games()
    agree = total = 0
    for g in games:
        a, n, first = play_on_device(g)
        agree += a
        total += n
        if first is not None:
            # the reference's own record: some selection up to this ply chose between two scores closer than 1e-5 (the network outputs
            # of the device and of torch differ by ~1e-7; W / N + c sqrt(ln Np / N) moves by that much)
            assert first['smallest_gap_so_far'] is not None and first['smallest_gap_so_far'] < 1e-5, (g['B'], first)
    assert total >= 500 and agree >= 1   # (the share itself is reported, not asserted: profiles/r06/real_net_agreement.txt)
