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

import os as _os


def _claim_hw_queues(wanted=8):
    """Lanes of games (rlzero_amd.selfplay) are HIP streams, and streams that share a hardware queue take turns: with HIP's default of
    4 queues per device a fourth lane lands in a queue that is already in use and the lanes serialise (512 games: 6.1 instead of 10.2 M
    simulations / s, profiles/r03/lane_sweeps.txt).  The runtime reads GPU_MAX_HW_QUEUES when it initialises, so the variable is set
    here -- on import, unless the caller has set it or the process has touched the GPU already -- and the number that will hold is
    remembered for plan_lanes()."""
    started = False
    try:   # the ROCm runtime holds /dev/kfd open from its first call on (torch.cuda.is_available() is such a call)
        started = any(_os.path.realpath('/proc/self/fd/' + fd) == '/dev/kfd' for fd in _os.listdir('/proc/self/fd'))
    except OSError:
        pass
    if 'GPU_MAX_HW_QUEUES' in _os.environ:
        try:
            return int(_os.environ['GPU_MAX_HW_QUEUES'])
        except ValueError:
            return 4
    if started:
        return 4  # too late: the runtime has created its queues
    _os.environ['GPU_MAX_HW_QUEUES'] = str(wanted)
    return wanted


HW_QUEUES = _claim_hw_queues()

