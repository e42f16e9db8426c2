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
    pipe2.collect_selfplay_data(8)
    state, prob, z = pipe2.data_buffer[0]
    assert state.shape == (4, 6, 6) and prob.shape == (36, ) and z in (-1.0, 0.0, 1.0)
    assert len(pipe2.data_buffer) >= 8 * 8 * 7
    pipe2.policy_update()
