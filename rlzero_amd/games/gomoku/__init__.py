from .game import GameControl
from .gomoku_env import GomokuEnv

__all__ = ['GomokuEnv', 'GameControl']
