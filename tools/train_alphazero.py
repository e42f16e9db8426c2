#!/usr/bin/env python3
"""AlphaZero training loop for Gomoku -- this repository's twin of the reference script
tools/train_alphazero.py (same import lines :11-14, attribute names :21-57, method names
``get_equi_data / collect_selfplay_data / policy_update / policy_evaluate / run`` and stdout
lines :124-136,161,170), running on the MI355X engine.

Two ways to collect self-play data:
  * ``selfplay_games_in_flight == 0`` (default): the reference's flow -- one game at a time
    through ``GameControl.start_self_play`` and ``AlphaZeroPlayer`` (search on the GPU);
  * ``selfplay_games_in_flight  > 0``: that many games in lock-step per collection round and GPU
    (``rlzero_amd.selfplay.BatchedSelfPlay``), the mode the hardware is built for; ``policy_evaluate``'s games then
    run in lock-step as well (``rlzero_amd.evaluate.BatchedEvaluation``).

Several GPUs (``python tools/train_alphazero.py --gpus N ...`` starts one process per GPU itself, or run it under
``torch.distributed.run``): the games of a collection round are dealt to the ranks by id (game g -> rank g mod N, no
collective inside the search), ONE gather brings the finished trajectories to rank 0, which alone keeps the replay buffer
and runs ``policy_update`` / ``policy_evaluate`` / the checkpoints (train_alphazero.py:81-137,164-190), then ONE broadcast
hands every rank the new weights for its next round.  A game's trajectory depends on (seed, game id) only, so the data --
and with them the losses -- are those of the single-process run on the same ids.

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
from collections.abc import Sequence

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


class ReplayBuffer(Sequence):
    """The reference's ``deque(maxlen=buffer_size)`` of augmented samples (train_alphazero.py:32, 59-79, 88-90) with the 8 symmetries
    formed WHEN A SAMPLE IS READ instead of when a game arrives: entry 8 j + k is symmetry k of the j-th sample ever added, in
    get_equi_data's order and with its numpy operations -- value for value what ``deque.extend(get_equi_data(play_data))`` would
    hold (tests/test_train_script.py) -- but a collection round of 512 games (52 k positions, 420 k entries) costs the host a list
    append per game where forming every symmetry up front cost more than the round's self-play on the GPU, and a mini-batch reads 32.
    Sequence protocol (len, index, iteration: what ``random.sample`` and the tests use), ``maxlen``, ``extend`` (ready-made
    entries: the reference flow), ``extend_samples`` (a game's un-augmented samples)."""

    def __init__(self, maxlen, board_size):
        self.maxlen, self.size = int(maxlen), int(board_size)
        self._blocks = deque()   # ('raw', [entries]) or ('game', states [P,4,B,B], pi grids [P,B,B], z [P])
        self._counts = deque()   # entries each block contributes
        self._skip = 0           # entries of the FIRST block that have fallen out (maxlen)
        self._len = 0

    def __len__(self):
        return self._len

    def _trim(self):
        over = self._len - self.maxlen
        while over > 0:
            left = self._counts[0] - self._skip
            if over >= left:
                self._blocks.popleft()
                self._counts.popleft()
                self._skip = 0
                self._len -= left
                over -= left
            else:
                self._skip += over
                self._len -= over
                over = 0

    def extend(self, entries):
        entries = list(entries)
        if entries:
            self._blocks.append(('raw', entries))
            self._counts.append(len(entries))
            self._len += len(entries)
            self._trim()

    def extend_samples(self, play_data):
        """A game's (state, mcts_prob, z) samples, un-augmented: 8 entries each, formed on access."""
        play_data = list(play_data)
        if not play_data:
            return
        states = np.stack([np.asarray(s_) for s_, _, _ in play_data])
        grids = np.stack([np.asarray(p_).reshape(self.size, self.size) for _, p_, _ in play_data])
        self._blocks.append(('game', states, grids, [w for _, _, w in play_data]))
        self._counts.append(8 * len(play_data))
        self._len += 8 * len(play_data)
        self._trim()

    def _entry(self, block, i):
        if block[0] == 'raw':
            return block[1][i]
        _, states, grids, zs = block
        j, k = divmod(i, 8)
        quarter_turns, mirrored = k // 2 + 1, k % 2 == 1
        turned = np.rot90(states[j:j + 1], quarter_turns, axes=(2, 3))            # get_equi_data's operations on a stack of one
        pi_turned = np.rot90(grids[j:j + 1, ::-1, :], quarter_turns, axes=(1, 2))
        if mirrored:
            return (np.ascontiguousarray(turned[:, :, :, ::-1])[0], np.ascontiguousarray(pi_turned[:, :, ::-1][:, ::-1, :]).reshape(1, -1)[0], zs[j])
        return np.ascontiguousarray(turned)[0], np.ascontiguousarray(pi_turned[:, ::-1, :]).reshape(1, -1)[0], zs[j]

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self[i] for i in range(*index.indices(self._len))]
        if index < 0:
            index += self._len
        if not 0 <= index < self._len:
            raise IndexError('ReplayBuffer index out of range')
        index += self._skip
        for block, count in zip(self._blocks, self._counts):
            if index < count:
                return self._entry(block, index)
            index -= count
        raise IndexError('ReplayBuffer index out of range')

    def __iter__(self):
        first = True
        for block, count in zip(self._blocks, self._counts):
            for i in range(self._skip if first else 0, count):
                yield self._entry(block, i)
            first = False


class TrainPipeline:

    def __init__(self, board_size=6, n_in_row=4, n_playout=400, game_batch_num=64, check_freq=50,
                 selfplay_games_in_flight=0, buffer_size=None, seed=None):
        """``buffer_size``: length of the replay deque.  None = the reference's 1000 (train_alphazero.py:32) in the
        reference flow; in the batched mode (``selfplay_games_in_flight > 0``) None sizes it to hold ONE collection
        round (games in flight x board cells x 8 symmetries) -- a documented deviation: with the reference's 1000 a
        256-game round (~200 k augmented samples) would keep its last 1000 samples and drop > 99 % of what the GPU
        produced.  Pass 1000 to get the reference's number in either mode.  ``seed``: of the batched mode's move draws
        (uniforms keyed (seed, game id, ply)); None = drawn on rank 0.  Under a launcher (RANK / WORLD_SIZE set, or an
        initialised process group) the pipeline is one of the ranks: see the module docstring."""
        self.rank, self.world = self._init_ranks()
        if self.world > 1 and selfplay_games_in_flight <= 0:
            raise ValueError('several ranks share a collection round: selfplay_games_in_flight must be > 0 (games per GPU)')
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
                max(1000, selfplay_games_in_flight * self.world * board_size * board_size * 8)
        self.buffer_size = int(buffer_size)
        self.batch_size = 32
        # (the reference's deque of augmented samples, the symmetries formed on access: ReplayBuffer)
        self.data_buffer = ReplayBuffer(self.buffer_size, self.board_size)
        self.play_batch_size = 1
        self.epochs = 5
        self.kl_targ = 0.02
        self.check_freq = check_freq
        self.game_batch_num = game_batch_num
        self.best_win_ratio = 0.0
        self.device = self._pick_device()
        self.pure_mcts_playout_num = 100
        self.selfplay_games_in_flight = selfplay_games_in_flight
        self.alphazero_agent = AlphaZeroAgent(self.board_size, device=self.device)
        self.mcts_player = AlphaZeroPlayer(self.alphazero_agent.policy_value_fn, n_playout=self.n_playout,
                                           c_puct=self.c_puct, is_selfplay=True)
        self._batched = None
        self._duel, self._duel_key, self._evaluations = None, None, 0
        self._next_game_id = 0
        self._trace_rounds = 0
        self.selfplay_seed = self._agree_on(random.getrandbits(31) if seed is None else int(seed))
        if self.world > 1:   # every rank starts from rank 0's weights
            from rlzero.algorithms import broadcast_weights
            broadcast_weights(self.alphazero_agent.policy_value_net, src=0)

    # ------------------------------------------------------------------ ranks
    @staticmethod
    def _init_ranks():
        """-> (rank, world).  One process per GPU; the process group is RCCL (backend nccl) on GPUs, gloo on the CPU or when
        RZ_DIST_BACKEND says so (two ranks on one GPU: RCCL refuses that)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world <= 1:
            return 0, 1
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('RZ_DIST_BACKEND', 'nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            local = 0 if os.environ.get('RZ_DIST_SINGLE_DEVICE') == '1' else int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
        return dist.get_rank(), dist.get_world_size()

    def _pick_device(self):
        if not torch.cuda.is_available():
            return torch.device('cpu')
        if self.world == 1:
            return torch.device('cuda')
        local = 0 if os.environ.get('RZ_DIST_SINGLE_DEVICE') == '1' else int(os.environ.get('LOCAL_RANK', str(self.rank)))
        torch.cuda.set_device(local)
        return torch.device('cuda', local)

    def _agree_on(self, value):
        """rank 0's ``value`` (an int) on every rank."""
        if self.world == 1:
            return value
        import torch.distributed as dist
        on_gpu = dist.get_backend() == 'nccl'
        t = torch.tensor([value], dtype=torch.int64, device=self.device if on_gpu else 'cpu')
        dist.broadcast(t, src=0)
        return int(t.item())

    # ------------------------------------------------------------------ data
    def get_equi_data(self, play_data):
        """8-fold augmentation: [(state, mcts_prob, winner_z), ...] -> 8x as many, in the reference's order (per sample: for
        i = 1..4 the pair rotated by i quarter turns, then its left-right mirror; train_alphazero.py:59-79).  A game's samples are
        turned together -- numpy's rot90 / flip on the stacked arrays, the per-sample results of `_symmetries` (which
        tests/golden/g5_equi.npz pins) as views of them: a 256-game round is 200 k samples, and one numpy call per plane and
        sample was most of what the trainer's collection round cost the host."""
        play_data = list(play_data)
        if not play_data:
            return []
        size = self.board_size
        states = np.stack([np.asarray(s_) for s_, _, _ in play_data])                                   # [P, 4, B, B]
        grids = np.stack([np.asarray(p_).reshape(size, size) for _, p_, _ in play_data])                # [P, B, B]
        upside_down = grids[:, ::-1, :]                                                                 # np.flipud per sample
        out_s, out_p = [], []
        for quarter_turns in (1, 2, 3, 4):
            turned = np.rot90(states, quarter_turns, axes=(2, 3))
            pi_turned = np.rot90(upside_down, quarter_turns, axes=(1, 2))
            out_s.append(np.ascontiguousarray(turned))
            out_p.append(np.ascontiguousarray(pi_turned[:, ::-1, :]).reshape(len(play_data), -1))      # flipud, flatten
            out_s.append(np.ascontiguousarray(turned[:, :, :, ::-1]))                                    # fliplr of every plane
            out_p.append(np.ascontiguousarray(pi_turned[:, :, ::-1][:, ::-1, :]).reshape(len(play_data), -1))   # flipud(fliplr)
        extend_data = []
        for j, (_, _, winner) in enumerate(play_data):
            for k in range(8):
                extend_data.append((out_s[k][j], out_p[k][j], winner))
        return extend_data

    def _play_games(self, game_ids, on_finished=None):
        """The trajectories of ``game_ids`` (this rank's share of a round), games in lock-step on this rank's GPU.  ``on_finished``: see
        BatchedSelfPlay.run_device (one rank, the move step on the device: the round's samples are formed while the GPU plays on)."""
        from rlzero.algorithms import BatchedSelfPlay
        if self._batched is None:
            # one to four lanes of games, whichever fills the GPU better (selfplay.plan_lanes)
            self._batched = BatchedSelfPlay.for_network(
                self.alphazero_agent.policy_value_net, self.board_size, self.n_in_row,
                n_games=self.selfplay_games_in_flight, n_playout=self.n_playout, c_puct=self.c_puct,
                device=str(self.device), temperature=self.temperature, seed=self.selfplay_seed)
        self._batched.refresh_weights()   # (every lane's evaluator: the learner has stepped / new weights have arrived)
        # the move step on the device (rz_play_*: the host reads the games from a log behind the GPU); RZ_TRAIN_HOST_MOVES=1: the
        # host-driven loop -- the same trajectories either way (tests/test_device_moves.py)
        if os.environ.get('RZ_TRAIN_HOST_MOVES') == '1':
            trajs = self._batched.run(game_ids) if len(game_ids) else []
            if on_finished is not None and trajs:
                on_finished(trajs)
        else:
            trajs = self._batched.run_device(game_ids, on_finished=on_finished) if len(game_ids) else []
        if os.environ.get('RZ_TRAIN_TRACE'):
            self._trace_round(trajs)
        return trajs

    def _trace_round(self, trajs):
        """RZ_TRAIN_TRACE=<directory>: one JSON line per collection round and rank -- a digest of the torch parameters, a digest
        of what every lane's HIP evaluator answers on a fixed batch of positions (i.e. of the weights it was handed), and the
        round's games.  What the multi-rank tests compare: every rank must search with the weights rank 0 learned."""
        import hashlib
        import json
        net = self.alphazero_agent.policy_value_net
        digest = hashlib.sha1()
        for p in net.parameters():
            digest.update(p.detach().cpu().numpy().tobytes())
        rng = np.random.RandomState(7)
        obs = torch.from_numpy((rng.rand(8, 4, self.board_size, self.board_size) < 0.3).astype(np.float32)).to(self.device)
        answers = []
        for lane in self._batched.lanes:
            with torch.cuda.stream(lane.stream):
                if getattr(lane.evaluator.hip, 'algo', 'split_f16') == 'split_f16_fp8':
                    answers.append('position-fed-only')   # (the opt-in FP8 mode takes no float planes: nothing to digest here)
                    continue
                logp, value = lane.evaluator.hip.forward(obs)
            lane.stream.synchronize()
            answers.append(hashlib.sha1(logp.cpu().numpy().tobytes() + value.cpu().numpy().tobytes()).hexdigest())
        rec = {'rank': self.rank, 'round': self._trace_rounds, 'params': digest.hexdigest(), 'lanes': answers,
               'games': {str(t.game_id): t.moves for t in trajs}}
        self._trace_rounds += 1
        with open(os.path.join(os.environ['RZ_TRAIN_TRACE'], 'rank%d.jsonl' % self.rank), 'a') as f:
            f.write(json.dumps(rec) + '\n')

    def _collect_batched(self, n_games, consume=None):
        """One collection round: ``n_games`` games in all, game g played by rank g mod world (selfplay.shard_game_ids),
        one gather to rank 0 (pi as float32: what the learner consumes).  -> start_self_play's tuples on rank 0, in game
        id order; [] on the other ranks.  ``consume`` (one rank): called with every game's tuple IN GAME-ID ORDER as soon as the game
        and all games before it have ended -- while the GPU plays the others --; the tuples it took are not returned."""
        from rlzero.algorithms import gather_trajectories
        ids = range(self._next_game_id, self._next_game_id + n_games)
        self._next_game_id += n_games
        if self.world > 1:
            local = self._play_games([g for g in ids if g % self.world == self.rank])
            merged = gather_trajectories(local, self.board_size, self.n_in_row, dst=0, pi_dtype=np.float32)
            return [t.as_reference_tuple() for t in merged] if merged is not None else []
        # one rank: every finished game becomes start_self_play's tuple (planes from its move list) WHILE the others are played --
        # behind an idle GPU that was a tenth of a round; the round's order stays the game ids'
        ready, cursor = {}, [ids.start]

        def take(trajs):
            for t in trajs:
                ready[t.game_id] = t.as_reference_tuple()
            while consume is not None and cursor[0] in ready:   # (the replay buffer's order is the reference's: game by game)
                consume(ready.pop(cursor[0]))
                cursor[0] += 1
        local = self._play_games(list(ids), on_finished=take)
        rest = [t for t in sorted(local, key=lambda t: t.game_id) if t.game_id >= cursor[0] or consume is None]
        return [ready[t.game_id] if t.game_id in ready else t.as_reference_tuple() for t in rest]

    def collect_selfplay_data(self, n_games=1):
        """collect self-play data for training."""
        def consume(game):
            winner, play_data = game
            play_data = list(play_data)
            self.episode_len = len(play_data)
            if os.environ.get('RZ_TRAIN_EAGER_SYMMETRIES') == '1':   # (the eight symmetries of every sample up front, as before round 6)
                self.data_buffer.extend(self.get_equi_data(play_data))
            else:
                self.data_buffer.extend_samples(play_data)
        if self.selfplay_games_in_flight > 0:
            games = self._collect_batched(max(n_games, self.selfplay_games_in_flight * self.world), consume=consume)
        else:
            games = [self.game.start_self_play(self.mcts_player, temperature=self.temperature)
                     for _ in range(n_games)]
        for game in games:
            consume(game)

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

    def _evaluate_batched(self, n_games):
        """The ``n_games`` evaluation games in lock-step on this GPU (rlzero_amd.evaluate.BatchedEvaluation: per game what
        start_play does with the two players below) -> their winner ids."""
        from rlzero.algorithms import BatchedEvaluation
        key = (n_games, self.pure_mcts_playout_num)
        if self._duel is None or self._duel_key != key:
            if self._duel is not None:
                self._duel.close()
            self._duel = BatchedEvaluation.for_network(
                self.alphazero_agent.policy_value_net, self.board_size, self.n_in_row, n_games=n_games,
                n_playout=self.n_playout, rollout_playouts=self.pure_mcts_playout_num, c_puct=self.c_puct, rollout_c_puct=5,
                device=str(self.device), seed=self.selfplay_seed)
            self._duel_key = key
        self._duel.refresh_weights()
        self._duel.seed = (self.selfplay_seed + 0x9E37 * self._evaluations) & 0x7fffffff   # fresh draws every evaluation
        self._evaluations += 1
        return [r.winner for r in self._duel.run()]

    def policy_evaluate(self, n_games=10):
        """Play the current policy against the pure-MCTS opponent (monitoring only)."""
        if self.selfplay_games_in_flight > 0:
            win_cnt = defaultdict(int)
            for winner in self._evaluate_batched(n_games):
                win_cnt[winner] += 1
            win_ratio = 1.0 * (win_cnt[1] + 0.5 * win_cnt[-1]) / n_games
            print('num_playouts:{}, win: {}, lose: {}, tie:{}'.format(self.pure_mcts_playout_num, win_cnt[1],
                                                                      win_cnt[2], win_cnt[-1]))
            return win_ratio
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

    def _sync_weights(self):
        """After rank 0's policy_update: its parameters on every rank (one broadcast; nothing to do in one process)."""
        if self.world > 1:
            from rlzero.algorithms import broadcast_weights
            broadcast_weights(self.alphazero_agent.policy_value_net, src=0)

    def run(self):
        """run the training pipeline."""
        lead = self.rank == 0   # rank 0 holds the buffer, learns, evaluates, saves and prints; the others play their share
        try:
            for i in range(self.game_batch_num):
                self.collect_selfplay_data(self.play_batch_size)
                if lead:
                    print('batch i:{}, episode_len:{}'.format(i + 1, self.episode_len))
                    if len(self.data_buffer) > self.batch_size:
                        loss, entropy = self.policy_update()
                self._sync_weights()
                if lead and (i + 1) % self.check_freq == 0:
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


def launch_ranks(n):
    """`--gpus n` without a launcher: run this command line under torch.distributed.run as a CHILD process (this process
    never touches the GPU); returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    import argparse
    ap = argparse.ArgumentParser(description='AlphaZero training for Gomoku on MI355X (no arguments: the reference script\'s run)')
    ap.add_argument('--gpus', type=int, default=1, help='processes (one per GPU); > 1 without a launcher starts them itself')
    ap.add_argument('--board', type=int, default=6)
    ap.add_argument('--n-in-row', type=int, default=4)
    ap.add_argument('--playouts', type=int, default=400)
    ap.add_argument('--batches', type=int, default=64, help='game_batch_num')
    ap.add_argument('--check-freq', type=int, default=50)
    ap.add_argument('--games-in-flight', type=int, default=0, help='per GPU; 0 = the reference flow, one game at a time')
    ap.add_argument('--seed', type=int, default=None)
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if args.seed is not None:   # a reproducible run: initial weights, random.sample of the replay buffer, numpy draws
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    pipe = TrainPipeline(board_size=args.board, n_in_row=args.n_in_row, n_playout=args.playouts, game_batch_num=args.batches,
                         check_freq=args.check_freq, selfplay_games_in_flight=args.games_in_flight, seed=args.seed)
    pipe.run()
    if pipe.world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
