"""``rlzero`` -- the reference's package name, provided by ``rlzero_amd``.

Only the AlphaZero self-play path exists here (scope): ``rlzero.mcts``,
``rlzero.games`` and ``rlzero.algorithms`` ARE the MI355X-native modules of ``rlzero_amd``
(same module objects, registered under both names), so the reference's import lines
(tools/train_alphazero.py:11-14) keep working unchanged.  Everything else of the reference
(DouDizhu/DMC, CFR, Atari, Go) is out of scope and absent.
"""
import importlib
import sys

_MODULES = (
    'mcts', 'mcts.player', 'mcts.alphazero_mcts', 'mcts.rollout_mcts',
    'games', 'games.base_env', 'games.gomoku', 'games.gomoku.gomoku_env', 'games.gomoku.game',
    'games.gomoku.policy_value_net', 'games.gomoku.alphazero_agent',
    'algorithms',
)

for _name in _MODULES:
    try:
        _real = importlib.import_module('rlzero_amd.' + _name)
    except ModuleNotFoundError as _exc:  # a module that does not exist (yet) is simply absent
        if _exc.name != 'rlzero_amd.' + _name:
            raise
        continue
    sys.modules[__name__ + '.' + _name] = _real
    if '.' not in _name:
        globals()[_name] = _real
