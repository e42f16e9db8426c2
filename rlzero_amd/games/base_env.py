"""Abstract environment interface kept from the reference
(rlzero/games/base_env.py:7-33).  The reference derives it from ``gymnasium.Env``; nothing
on the AlphaZero path uses gymnasium, so it is optional here."""
import copy

try:  # pragma: no cover - gymnasium is not installed in the build image
    import gymnasium
    _Base = gymnasium.Env
except Exception:  # noqa: BLE001
    _Base = object


class Error(Exception):
    """Raised for bad board / start-player arguments (the reference borrows ``uu.Error``,
    gomoku_env.py:4, which no longer exists in Python 3.13)."""


class BaseEnv(_Base):

    def __init__(self):
        pass

    def render(self):
        raise NotImplementedError

    def current_player(self):
        raise NotImplementedError

    def legal_actions(self, player):
        raise NotImplementedError

    def returns(self):
        raise NotImplementedError

    def clone(self):
        return copy.deepcopy(self)

    def is_terminal(self):
        raise NotImplementedError
