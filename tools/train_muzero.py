#!/usr/bin/env python3
"""MuZero on CartPole-v1 (BASELINE.json configs[4]): batched self-play on the GPU
(HIP search-tree kernels + the learned model on PyTorch-ROCm), replay buffer, K = 5 unrolled training.

The reference repository names MuZero (README.md:3) but has no script for it; this one follows the layout of
its ``tools/train_alphazero.py`` (a pipeline class with ``collect_selfplay_data`` / ``policy_update`` / ``run``).

    python tools/train_muzero.py --envs 256 --iterations 60
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


class MuZeroPipeline(object):

    def __init__(self, n_envs=256, n_sims=50, unroll_steps=5, td_steps=10, discount=0.997, batch_size=256,
                 moves_per_iteration=20, updates_per_iteration=40, device='cuda:0', seed=0):
        from rlzero_amd.muzero import CartPoleBatch, MuZeroAgent, MuZeroSelfPlay, ReplayBuffer
        self.agent = MuZeroAgent(device=device)
        self.env = CartPoleBatch(n_envs, device, seed=seed)
        self.selfplay = MuZeroSelfPlay(self.agent.net, self.env, n_sims=n_sims, discount=discount, seed=seed)
        self.buffer = ReplayBuffer(unroll_steps=unroll_steps, td_steps=td_steps, discount=discount, seed=seed)
        self.batch_size = batch_size
        self.moves_per_iteration = moves_per_iteration
        self.updates_per_iteration = updates_per_iteration
        self.episode_len = 0.0

    def collect_selfplay_data(self):
        episodes = self.selfplay.collect(self.moves_per_iteration)
        for ep in episodes:
            self.buffer.add(ep)
        if episodes:
            self.episode_len = float(np.mean([len(ep) for ep in episodes]))
        return len(episodes)

    def policy_update(self):
        out = (0.0, 0.0, 0.0, 0.0)
        for _ in range(self.updates_per_iteration):
            out = self.agent.learn(self.buffer.sample(self.batch_size, self.env.n_actions))
        return out

    def run(self, iterations):
        for i in range(iterations):
            n = self.collect_selfplay_data()
            if len(self.buffer) < 8:
                continue
            loss, lv, lr_, lp = self.policy_update()
            print('batch i:{}, episodes:{}, episode_len:{:.1f}, loss:{:.4f}, value:{:.4f}, reward:{:.4f}, '
                  'policy:{:.4f}'.format(i + 1, n, self.episode_len, loss, lv, lr_, lp), flush=True)
        return self.episode_len


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=256)
    ap.add_argument('--sims', type=int, default=50)
    ap.add_argument('--iterations', type=int, default=60)
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--save', default='')
    args = ap.parse_args()
    pipe = MuZeroPipeline(n_envs=args.envs, n_sims=args.sims, device=args.device)
    pipe.run(args.iterations)
    if args.save:
        pipe.agent.save_model(args.save)
