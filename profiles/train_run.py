#!/usr/bin/env python3
"""Runs this repository's twin of the reference trainer (tools/train_alphazero.py) on the GPU in both collection
modes and records its stdout with wall-clock per batch: python3 profiles/train_run.py > profiles/r01/train_alphazero_run.log"""
import importlib.util
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('train_alphazero', os.path.join(REPO, 'tools', 'train_alphazero.py'))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
os.chdir(tempfile.mkdtemp())
torch.manual_seed(0)
np.random.seed(0)

print('== reference flow: one game at a time through GameControl.start_self_play / AlphaZeroPlayer (6x6, 4 in a row, 400 playouts) ==')
pipe = mod.TrainPipeline(board_size=6, n_in_row=4, n_playout=400, game_batch_num=6, check_freq=6)
pipe.pure_mcts_playout_num = 200
t0 = time.perf_counter()
pipe.run()
dt = time.perf_counter() - t0
print('-- 6 batches (6 games, 6 updates, 1 evaluation of 10 games vs RolloutPlayer) in %.1f s' % dt)

print('== batched collection: 512 games in flight (15x15, 5 in a row, 800 playouts: BASELINE configs[3], one resident lane) ==')
G = int(os.environ.get('RZ_TRAIN_GAMES', '512'))
pipe = mod.TrainPipeline(board_size=15, n_in_row=5, n_playout=800, game_batch_num=3, check_freq=1000,
                         selfplay_games_in_flight=G)
spent = {'play': 0.0}
real_play = pipe._play_games


def timed_play(ids, **kw):
    t = time.perf_counter()
    out = real_play(ids, **kw)
    spent['play'] += time.perf_counter() - t
    return out


pipe._play_games = timed_play
for i, n in enumerate((G, G, 4 * G, 8 * G, 16 * G)):
    spent['play'] = 0.0
    t0 = time.perf_counter()
    pipe.collect_selfplay_data(n)
    t1 = time.perf_counter()
    loss, entropy = pipe.policy_update()
    t2 = time.perf_counter()
    n_pos = len(pipe.data_buffer) // 8  # the reference's deque(maxlen=1000) keeps the newest 1000 samples
    print('-- round %d: %d games (%d in flight) collected in %.2f s (%.1f games/s): the self-play call %.2f s (finished games become samples inside it, while the GPU plays), the rest '
          '(replay buffer) %.2f s; buffer %d positions x 8 symmetries, update %.2f s, loss %.4f entropy %.4f' % (
              i + 1, n, G, t1 - t0, n / (t1 - t0), spent['play'], t1 - t0 - spent['play'], n_pos, t2 - t1, loss, entropy))
print('(a round of exactly as many games as slots ends with its LONGEST game -- up to 225 plies of ~17 ms where the mean game has 102 --;\n'
      ' a round of several times the slots refills them from the queue and runs at the engine\'s steady rate: bench.py\'s selfplay leg)')
