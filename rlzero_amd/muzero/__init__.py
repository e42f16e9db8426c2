"""MuZero on MI355X (BASELINE.json configs[4]: learned dynamics on CartPole-v1, 50 simulations per move,
recurrent unroll K = 5).  The reference names MuZero but ships none (README.md:3,
rlzero/algorithms/rl_args.py:21-24), so this package is build-defined: the published algorithm
(arXiv:1911.08265v2) with the search tree in hand-written HIP kernels (csrc/rz_muzero.hip, C ABI rz_mz_*),
the small MLPs of the learned model and the learner on PyTorch-ROCm."""
from .agent import MuZeroAgent
from .cartpole import CartPoleBatch
from .network import MuZeroNet
from .selfplay import MuZeroSelfPlay, ReplayBuffer
from .tree import MuZeroTree

__all__ = ['MuZeroAgent', 'CartPoleBatch', 'MuZeroNet', 'MuZeroSelfPlay', 'ReplayBuffer', 'MuZeroTree']
