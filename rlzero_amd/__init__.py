"""rlzero_amd -- MI355X-native AlphaZero self-play MCTS behind the RLZero Python API.

Scope: ONE hot path of jianzhnie/RLZero (select -> expand -> evaluate -> backup, Gomoku /
TicTacToe rules, batched policy+value forward), see DESIGN.md.  The package mirrors the
reference's module layout for that path (``rlzero_amd.mcts``, ``rlzero_amd.games``); the
top-level ``rlzero`` package of this repository aliases those modules so the reference's
import lines (tools/train_alphazero.py:11-14) work unchanged.

The tree / rules kernels live in ``csrc/`` (HIP, gfx950) behind the C ABI declared in
``include/rlzero_hip.h``; there is no CPU fallback.
"""
__version__ = '0.1.0'
