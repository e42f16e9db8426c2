"""Trainer API of the AlphaZero path.

In the reference the AlphaZero trainer is the ``TrainPipeline`` class inside the script
tools/train_alphazero.py:17-190 (``rlzero/algorithms`` only holds the unrelated DMC / CFR
code).  The batched, multi-GPU self-play collector that replaces its sequential
``collect_selfplay_data`` loop is ``rlzero_amd.selfplay``; it is re-exported here under the
name BASELINE.json uses, with the lock-step counterpart of ``policy_evaluate``'s games
(``rlzero_amd.evaluate``).
"""
from ..evaluate import BatchedEvaluation, DuelResult
from ..selfplay import (BatchedSelfPlay, Trajectory, broadcast_weights, gather_trajectories, shard_game_ids)

__all__ = ['BatchedSelfPlay', 'BatchedEvaluation', 'DuelResult', 'Trajectory', 'gather_trajectories', 'shard_game_ids', 'broadcast_weights']
