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


def _runtime_started():
    """The ROCm runtime holds /dev/kfd open from its first call on (torch.cuda.is_available() is such a call)."""
    try:
        return any(_os.path.realpath('/proc/self/fd/' + fd) == '/dev/kfd' for fd in _os.listdir('/proc/self/fd'))
    except OSError:
        return False


def _claim_hw_queues(wanted=8):
    """Lanes of games (rlzero_amd.selfplay) are HIP streams, and streams that share a hardware queue take turns: with HIP's default of
    4 queues per device a fourth lane lands in a queue that is already in use and the lanes serialise (512 games: 6.1 instead of 10.2 M
    simulations / s, profiles/r03/lane_sweeps.txt).  The runtime reads GPU_MAX_HW_QUEUES when it initialises, so the variable is set
    here -- on import, unless the caller has set it (GPU_MAX_HW_QUEUES, or RZ_HW_QUEUES for this package alone) or the process has
    touched the GPU already -- and the number that will hold is remembered for plan_lanes().  -> (queues, too_late)."""
    if 'GPU_MAX_HW_QUEUES' in _os.environ:
        try:
            return int(_os.environ['GPU_MAX_HW_QUEUES']), False
        except ValueError:
            return 4, False
    if _runtime_started():
        return 4, True   # too late: the runtime has created its queues
    try:
        wanted = int(_os.environ.get('RZ_HW_QUEUES', wanted))
    except ValueError:
        pass
    _os.environ['GPU_MAX_HW_QUEUES'] = str(wanted)
    return wanted, False


HW_QUEUES, HW_QUEUES_TOO_LATE = _claim_hw_queues()


def configure(hw_queues=8):
    """The explicit form of what importing this package does: ask the HIP runtime for ``hw_queues`` hardware queues per device
    (every lane of games needs one of its own; four lanes -- the layout of 512 games per GPU -- need 8).  Must run before the
    process's first GPU call: the runtime reads GPU_MAX_HW_QUEUES once, when it starts.  Returns the number that holds; if the
    runtime is up already nothing can change, a RuntimeWarning says so and plan_lanes() will use fewer lanes.  The same knob
    without code: ``RZ_HW_QUEUES=<n>`` (read at import) or ``GPU_MAX_HW_QUEUES=<n>`` (the runtime's own variable) in the
    environment."""
    global HW_QUEUES, HW_QUEUES_TOO_LATE
    import warnings
    hw_queues = int(hw_queues)
    if _runtime_started():
        if hw_queues != HW_QUEUES:
            HW_QUEUES_TOO_LATE = True
            warnings.warn('rlzero_amd.configure(hw_queues=%d): the HIP runtime of this process has started with %d hardware queues; '
                          'import rlzero_amd (or call configure) before the first GPU call' % (hw_queues, HW_QUEUES), RuntimeWarning,
                          stacklevel=2)
        return HW_QUEUES
    _os.environ['GPU_MAX_HW_QUEUES'] = str(hw_queues)
    HW_QUEUES, HW_QUEUES_TOO_LATE = hw_queues, False
    return HW_QUEUES
