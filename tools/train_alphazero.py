#!/usr/bin/env python3
"""AlphaZero training loop for Gomoku -- this repository's twin of the reference script
tools/train_alphazero.py (same import lines :11-14, attribute names :21-57, method names
``get_equi_data / collect_selfplay_data / policy_update / policy_evaluate / run`` and stdout
lines :124-136,161,170), running on the MI355X engine.

Two ways to collect self-play data:
  * ``selfplay_games_in_flight == 0`` (default): the reference's flow -- one game at a time
    through ``GameControl.start_self_play`` and ``AlphaZeroPlayer`` (search on the GPU);
  * ``selfplay_games_in_flight  > 0``: that many games in lock-step per collection round
    (``rlzero_amd.selfplay.BatchedSelfPlay``), the mode the hardware is built for.

Reference quirks kept on purpose: ``lr_multiplier`` is adapted but never
applied and ``learn_rate`` is never passed to the agent (D-7); ``policy_evaluate`` counts
``win_cnt[1]`` as wins although player ids are 0/1 (D-6); ``get_equi_data`` rotates the planes
by +i*90 degrees and pi by -i*90 degrees for odd i (D-9).
"""
from __future__ import print_function

import os
import random
import sys
from collections import defaultdict, deque

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from rlzero.games.gomoku import GameControl, GomokuEnv
from rlzero.games.gomoku.alphazero_agent import AlphaZeroAgent
from rlzero.mcts.alphazero_mcts import AlphaZeroPlayer
from rlzero.mcts.rollout_mcts import RolloutPlayer


def _symmetries(planes, pi_grid):
    """The 8 (state, pi) pairs the reference generates for one sample, in its order: for
    i = 1..4 the pair rotated by i quarter turns, then its left-right mirror.  The policy goes
    through the reference's flipud / rot90 / flipud sandwich (train_alphazero.py:65-78)."""
    out = []
    upside_down = np.flipud(pi_grid)
    for quarter_turns in (1, 2, 3, 4):
        turned = np.array([np.rot90(plane, quarter_turns) for plane in planes])
        pi_turned = np.rot90(upside_down, quarter_turns)
        out.append((turned, np.flipud(pi_turned).flatten()))
        mirrored = np.array([np.fliplr(plane) for plane in turned])
        out.append((mirrored, np.flipud(np.fliplr(pi_turned)).flatten()))
    return out


class TrainPipeline:

    def __init__(self, board_size=6, n_in_row=4, n_playout=400, game_batch_num=64, check_freq=50,
                 selfplay_games_in_flight=0, buffer_size=None):
        """``buffer_size``: length of the replay deque.  None = the reference's 1000 (train_alphazero.py:32) in the
        reference flow; in the batched mode (``selfplay_games_in_flight > 0``) None sizes it to hold ONE collection
        round (games in flight x board cells x 8 symmetries) -- a documented deviation: with the reference's 1000 a
        256-game round (~200 k augmented samples) would keep its last 1000 samples and drop > 99 % of what the GPU
        produced.  Pass 1000 to get the reference's number in either mode."""
        # board and game
        self.board_size = board_size
        self.n_in_row = n_in_row
        self.board = GomokuEnv(board_size=self.board_size, n_in_row=self.n_in_row)
        self.game = GameControl(self.board)
        # training hyper-parameters (train_alphazero.py:26-41)
        self.learn_rate = 2e-3
        self.lr_multiplier = 1.0
        self.temperature = 1.0
        self.n_playout = n_playout
        self.c_puct = 5
        if buffer_size is None:
            buffer_size = 1000 if selfplay_games_in_flight <= 0 else \
                max(1000, selfplay_games_in_flight * board_size * board_size * 8)
        self.buffer_size = int(buffer_size)
        self.batch_size = 32
        self.data_buffer = deque(maxlen=self.buffer_size)
        self.play_batch_size = 1
        self.epochs = 5
        self.kl_targ = 0.02
        self.check_freq = check_freq
        self.game_batch_num = game_batch_num
        self.best_win_ratio = 0.0
        self.device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        self.pure_mcts_playout_num = 100
        self.selfplay_games_in_flight = selfplay_games_in_flight
        self.alphazero_agent = AlphaZeroAgent(self.board_size, device=self.device)
        self.mcts_player = AlphaZeroPlayer(self.alphazero_agent.policy_value_fn, n_playout=self.n_playout,
                                           c_puct=self.c_puct, is_selfplay=True)
        self._batched = None
        self._next_game_id = 0

    # ------------------------------------------------------------------ data
    def get_equi_data(self, play_data):
        """8-fold augmentation: [(state, mcts_prob, winner_z), ...] -> 8x as many."""
        size = self.board_size
        extend_data = []
        for state, mcts_prob, winner in play_data:
            for equi_state, equi_prob in _symmetries(state, mcts_prob.reshape(size, size)):
                extend_data.append((equi_state, equi_prob, winner))
        return extend_data

    def _collect_batched(self, n_games):
        from rlzero.algorithms import BatchedSelfPlay
        if self._batched is None:
            # one or two lanes of games, whichever fills the GPU better (selfplay.plan_lanes)
            self._batched = BatchedSelfPlay.for_network(
                self.alphazero_agent.policy_value_net, self.board_size, self.n_in_row,
                n_games=self.selfplay_games_in_flight, n_playout=self.n_playout, c_puct=self.c_puct,
                device=str(self.device), temperature=self.temperature, seed=random.getrandbits(31))
        self._batched.refresh_weights()
        ids = range(self._next_game_id, self._next_game_id + n_games)
        self._next_game_id += n_games
        return [t.as_reference_tuple() for t in self._batched.run(ids)]

    def collect_selfplay_data(self, n_games=1):
        """collect self-play data for training."""
        if self.selfplay_games_in_flight > 0:
            games = self._collect_batched(max(n_games, self.selfplay_games_in_flight))
        else:
            games = [self.game.start_self_play(self.mcts_player, temperature=self.temperature)
                     for _ in range(n_games)]
        for winner, play_data in games:
            play_data = list(play_data)
            self.episode_len = len(play_data)
            self.data_buffer.extend(self.get_equi_data(play_data))

    # ------------------------------------------------------------------ learning
    def policy_update(self):
        """update the policy-value net."""
        mini_batch = random.sample(self.data_buffer, self.batch_size)
        state_batch, mcts_probs_batch, winner_batch = (list(col) for col in zip(*mini_batch))
        old_probs, old_v = self.alphazero_agent.policy_value(state_batch)
        for _ in range(self.epochs):
            loss, entropy = self.alphazero_agent.learn(state_batch, mcts_probs_batch, winner_batch)
            new_probs, new_v = self.alphazero_agent.policy_value(state_batch)
            kl = np.mean(np.sum(old_probs * (np.log(old_probs + 1e-10) - np.log(new_probs + 1e-10)), axis=1))
            if kl > self.kl_targ * 4:  # early stopping if D_KL diverges badly
                break
        if kl > self.kl_targ * 2 and self.lr_multiplier > 0.1:
            self.lr_multiplier /= 1.5
        elif kl < self.kl_targ / 2 and self.lr_multiplier < 10:
            self.lr_multiplier *= 1.5
        z = np.array(winner_batch)
        explained_var_old = 1 - np.var(z - old_v.flatten()) / np.var(z)
        explained_var_new = 1 - np.var(z - new_v.flatten()) / np.var(z)
        print(('kl:{:.5f},'
               'lr_multiplier:{:.3f},'
               'loss:{},'
               'entropy:{},'
               'explained_var_old:{:.3f},'
               'explained_var_new:{:.3f}').format(kl, self.lr_multiplier, loss, entropy, explained_var_old,
                                                  explained_var_new))
        return loss, entropy

    def policy_evaluate(self, n_games=10):
        """Play the current policy against the pure-MCTS opponent (monitoring only)."""
        current_mcts_player = AlphaZeroPlayer(self.alphazero_agent.policy_value_fn, n_playout=self.n_playout,
                                              c_puct=self.c_puct)
        pure_mcts_player = RolloutPlayer(n_playout=self.pure_mcts_playout_num, c_puct=5)
        win_cnt = defaultdict(int)
        for i in range(n_games):
            winner = self.game.start_play(current_mcts_player, pure_mcts_player, start_player=i % 2, is_shown=0)
            win_cnt[winner] += 1
        win_ratio = 1.0 * (win_cnt[1] + 0.5 * win_cnt[-1]) / n_games
        print('num_playouts:{}, win: {}, lose: {}, tie:{}'.format(self.pure_mcts_playout_num, win_cnt[1],
                                                                  win_cnt[2], win_cnt[-1]))
        return win_ratio

    def run(self):
        """run the training pipeline."""
        try:
            for i in range(self.game_batch_num):
                self.collect_selfplay_data(self.play_batch_size)
                print('batch i:{}, episode_len:{}'.format(i + 1, self.episode_len))
                if len(self.data_buffer) > self.batch_size:
                    loss, entropy = self.policy_update()
                if (i + 1) % self.check_freq == 0:
                    print('current self-play batch: {}'.format(i + 1))
                    win_ratio = self.policy_evaluate()
                    self.alphazero_agent.save_model('./current_policy.model')
                    if win_ratio > self.best_win_ratio:
                        print('New best policy!!!!!!!!')
                        self.best_win_ratio = win_ratio
                        self.alphazero_agent.save_model('./best_policy.model')
                        if self.best_win_ratio == 1.0 and self.pure_mcts_playout_num < 5000:
                            self.pure_mcts_playout_num += 1000
                            self.best_win_ratio = 0.0
        except KeyboardInterrupt:
            print('\n\rquit')


if __name__ == '__main__':
    training_pipeline = TrainPipeline()
    training_pipeline.run()
