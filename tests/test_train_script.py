"""tools/train_alphazero.py (this repository's twin of the reference's trainer script)."""
import importlib.util
import os
import types

import numpy as np
import pytest
from conftest import REPO


def _load():
    spec = importlib.util.spec_from_file_location('train_alphazero', os.path.join(REPO, 'tools', 'train_alphazero.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_get_equi_data_matches_reference(g5):
    """The +i / -i rotation quirk of the reference's augmentation, output for output."""
    mod = _load()
    fake = types.SimpleNamespace(board_size=4)
    out = mod.TrainPipeline.get_equi_data(fake, [(g5['state'], g5['pi'], 1.0)])
    assert len(out) == 8
    assert np.array_equal(np.array([s for s, _, _ in out]), g5['equi_states'])
    assert np.array_equal(np.array([p for _, p, _ in out]), g5['equi_pis'])
    assert np.array_equal(np.array([z for _, _, z in out]), g5['equi_z'])


def test_replay_buffer_is_the_reference_s_deque_of_augmented_samples(g5):
    """ReplayBuffer forms the 8 symmetries of a sample when it is read: entry for entry -- values, order, the maxlen window (also
    through the middle of a game), negative and sliced indices, ready-made entries beside lazy ones -- it is
    deque(maxlen).extend(get_equi_data(play_data)) (train_alphazero.py:32, 59-79, 88-90), and random.sample draws the same batch."""
    import random
    from collections import deque
    mod = _load()
    rng = np.random.RandomState(3)
    B = 4
    fake = types.SimpleNamespace(board_size=B)
    games = [[(g5['state'], g5['pi'], 1.0)]]
    for n in (5, 1, 9, 3):
        games.append([((rng.rand(4, B, B) < 0.4).astype(np.float64), rng.dirichlet(np.ones(B * B)), float(rng.choice([-1.0, 0.0, 1.0]))) for _ in range(n)])
    for maxlen in (1000, 100, 37, 8):
        ref, buf = deque(maxlen=maxlen), mod.ReplayBuffer(maxlen, B)
        for i, game in enumerate(games):
            ref.extend(mod.TrainPipeline.get_equi_data(fake, game))
            if i == 2:
                buf.extend(mod.TrainPipeline.get_equi_data(fake, game))   # ready-made entries (the reference flow's path)
            else:
                buf.extend_samples(game)
            assert len(buf) == len(ref) and buf.maxlen == maxlen
            for (s1, p1, z1), (s2, p2, z2) in zip(buf, ref):
                assert np.array_equal(s1, s2) and np.array_equal(p1, p2) and z1 == z2 and s1.shape == (4, B, B) and p1.shape == (B * B, )
        for i in (0, -1, len(ref) // 2):
            assert np.array_equal(buf[i][0], ref[i][0]) and np.array_equal(buf[i][1], ref[i][1]) and buf[i][2] == ref[i][2]
        assert len(buf[2:5]) == min(3, max(0, len(ref) - 2))
        with pytest.raises(IndexError):
            buf[len(ref)]
        if len(ref) >= 8:
            random.seed(5)
            a = random.sample(buf, 8)
            random.seed(5)
            b = random.sample(list(ref), 8)
            assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] for x, y in zip(a, b))


def test_reference_import_lines_and_names():
    text = open(os.path.join(REPO, 'tools', 'train_alphazero.py')).read()
    for line in ('from rlzero.games.gomoku import GameControl, GomokuEnv',
                 'from rlzero.games.gomoku.alphazero_agent import AlphaZeroAgent',
                 'from rlzero.mcts.alphazero_mcts import AlphaZeroPlayer',
                 'from rlzero.mcts.rollout_mcts import RolloutPlayer'):
        assert line in text
    mod = _load()
    for name in ('get_equi_data', 'collect_selfplay_data', 'policy_update', 'policy_evaluate', 'run'):
        assert callable(getattr(mod.TrainPipeline, name))


def _numbers(line):
    """'kl:0.00087,lr_multiplier:1.500,loss:4.55,...' -> {name: float}"""
    return {k: float(v) for k, v in (item.split(':') for item in line.split(','))}


def _run_g7(mod, g7, pipe, capsys, tol):
    """pipe.run() of this repository's twin against the reference's captured run: same episode_len / buffer
    length per batch, same log lines, every number policy_update prints within ``tol``."""
    import random
    import torch
    from conftest import unhex
    from oracle.evaluators import numpy_weights
    torch.manual_seed(g7['seed'])
    np.random.seed(g7['seed'])
    random.seed(g7['seed'])
    pipe.alphazero_agent.policy_value_net.load_state_dict(
        {k: torch.from_numpy(v) for k, v in numpy_weights(g7['B'], g7['weight_seed']).items()})
    losses = []
    real_update = pipe.policy_update
    pipe.policy_update = lambda: losses.append(real_update()) or losses[-1]
    seen = []
    real_collect = pipe.collect_selfplay_data

    def collect(n_games=1):
        real_collect(n_games)
        seen.append((pipe.episode_len, len(pipe.data_buffer)))

    pipe.collect_selfplay_data = collect
    capsys.readouterr()
    pipe.run()
    out = capsys.readouterr().out.splitlines()
    want = g7['stdout']
    assert seen == [(b['episode_len'], b['buffer_len']) for b in g7['batches']]
    assert len(out) == len(want)
    for got_line, want_line in zip(out, want):
        if want_line.startswith('batch i:'):
            assert got_line == want_line
        else:
            a, b = _numbers(got_line), _numbers(want_line)
            assert list(a) == list(b)
            printed = {'kl': 1e-5, 'lr_multiplier': 0.0, 'explained_var_old': 1e-3, 'explained_var_new': 1e-3}
            for k in a:  # loss / entropy are printed in full; the others rounded to 5 / 3 decimals
                assert abs(a[k] - b[k]) <= 10 * tol + printed.get(k, 0.0), (k, a[k], b[k])
    for (loss, entropy), b in zip(losses, g7['batches']):
        assert abs(loss - unhex(b['loss'])) <= tol and abs(entropy - unhex(b['entropy'])) <= tol
    assert pipe.lr_multiplier == unhex(g7['batches'][-1]['lr_multiplier'])
    sd = pipe.alphazero_agent.policy_value_net.state_dict()
    for k, v in g7['final_weights_abs_sum'].items():
        assert abs(float(sd[k].double().abs().sum()) - unhex(v)) <= 100 * tol * max(1.0, unhex(v)), k


def test_training_run_matches_the_reference_on_cpu(g7, capsys, monkeypatch, tmp_path):
    """The reference's 4-batch TrainPipeline.run() (tests/golden/g7_train.json.gz, captured from the reference itself)
    against this repository's trainer twin + AlphaZeroAgent.learn on the CPU.  The games come from the oracle driven
    by the fixture's uniforms (the same vlin search the reference ran; the HIP search is compared with it in the gpu
    twin of this test), everything behind them -- get_equi_data, the deque, random.sample, policy_update, learn, the
    log lines -- is the product's code: loss / entropy to 1e-5 (train_alphazero.py:92-137,170; alphazero_agent.py:59-86)."""
    import torch
    from conftest import unhex
    from oracle import evaluators as ev
    from oracle.gomoku_ref import RefGomoku
    from oracle.mcts_ref import RefPlayer, inverse_cdf_choice, self_play_game
    torch.set_num_threads(1)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)  # the trainer picks its device from this
    mod = _load()
    pipe = mod.TrainPipeline(board_size=g7['B'], n_in_row=g7['n'], n_playout=g7['n_playout'],
                             game_batch_num=len(g7['batches']), check_freq=50)
    choice = inverse_cdf_choice([unhex(u) for u in g7['u']])
    moves_seen = []

    def oracle_self_play(player, temperature=1e-3):
        ref_player = RefPlayer(ev.vlin, g7['n_playout'], g7['c_puct'], is_selfplay=True, choice=choice)
        winner, data, moves = self_play_game(RefGomoku(g7['B'], g7['n']), ref_player, temperature=temperature)
        moves_seen.append(moves)
        return winner, data

    pipe.game.start_self_play = oracle_self_play
    _run_g7(mod, g7, pipe, capsys, 1e-5)
    assert moves_seen == [b['moves'] for b in g7['batches']]
    assert len(choice.used) == len(g7['u'])


@pytest.mark.gpu
def test_training_run_matches_the_reference_on_gpu(g7, capsys, monkeypatch, tmp_path):
    """The same captured reference run against the WHOLE product on the MI355X: self-play through
    GameControl.start_self_play / AlphaZeroPlayer on the HIP engine (vlin as a host policy_value_fn, the reference's
    np.random.choice fed the fixture's uniforms), learner on the GPU: identical games and buffer, loss / entropy within
    1e-4 of the reference's CPU numbers (20 Adam steps on another device's convolution kernels)."""
    from conftest import unhex
    from oracle import evaluators as ev
    from rlzero.mcts.alphazero_mcts import AlphaZeroPlayer
    monkeypatch.chdir(tmp_path)
    mod = _load()
    pipe = mod.TrainPipeline(board_size=g7['B'], n_in_row=g7['n'], n_playout=g7['n_playout'],
                             game_batch_num=len(g7['batches']), check_freq=50)
    assert str(pipe.device).startswith('cuda')
    pipe.mcts_player = AlphaZeroPlayer(ev.vlin, n_playout=g7['n_playout'], c_puct=g7['c_puct'], is_selfplay=True)
    us = iter([unhex(u) for u in g7['u']])

    def choice(acts, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return np.asarray(acts)[cdf.searchsorted(next(us), side='right')]

    monkeypatch.setattr(np.random, 'choice', choice)
    games = []
    real = pipe.game.start_self_play

    def spy(player, temperature=1e-3):
        winner, data = real(player, temperature=temperature)
        games.append([int(m) for m in pipe.board.states.keys()])
        return winner, data

    pipe.game.start_self_play = spy
    _run_g7(mod, g7, pipe, capsys, 1e-4)
    assert games == [b['moves'] for b in g7['batches']]
    assert next(us, None) is None
    pipe.mcts_player.mcts._engine.close()


@pytest.mark.gpu
def test_train_pipeline_runs_on_gpu(tmp_path, capsys, monkeypatch):
    """Two batches of the reference flow and one batched collection: data flows from the GPU
    search into the learner, evaluation against RolloutPlayer runs, checkpoints are written."""
    import torch
    monkeypatch.chdir(tmp_path)
    mod = _load()
    torch.manual_seed(0)
    np.random.seed(0)
    pipe = mod.TrainPipeline(board_size=6, n_in_row=4, n_playout=40, game_batch_num=2, check_freq=2)
    pipe.pure_mcts_playout_num = 20
    pipe.batch_size = 16
    pipe.run()
    out = capsys.readouterr().out
    assert 'batch i:1, episode_len:' in out and 'kl:' in out and 'num_playouts:20, win:' in out
    assert os.path.isfile(tmp_path / 'current_policy.model' / 'model.th')
    assert len(pipe.data_buffer) >= 8 * 7
    # batched collection: 8 games in flight feed the same buffer format
    pipe2 = mod.TrainPipeline(board_size=6, n_in_row=4, n_playout=30, game_batch_num=1, check_freq=50,
                              selfplay_games_in_flight=8)
    assert pipe2.buffer_size == 8 * 36 * 8 and pipe.buffer_size == 1000  # one round of the batched mode / the reference's
    assert mod.TrainPipeline(selfplay_games_in_flight=8, buffer_size=1000).data_buffer.maxlen == 1000
    pipe2.collect_selfplay_data(8)
    state, prob, z = pipe2.data_buffer[0]
    assert state.shape == (4, 6, 6) and prob.shape == (36, ) and z in (-1.0, 0.0, 1.0)
    assert len(pipe2.data_buffer) >= 8 * 8 * 7
    pipe2.policy_update()
    # ... and its evaluation games run in lock-step too (rlzero_amd.evaluate), reported with the reference's line
    pipe2.pure_mcts_playout_num = 20
    capsys.readouterr()
    ratio = pipe2.policy_evaluate(4)
    assert 0.0 <= ratio <= 1.0 and 'num_playouts:20, win:' in capsys.readouterr().out
    assert pipe2._duel.n_slots == 4 and pipe2._duel.rollout_playouts == 20
