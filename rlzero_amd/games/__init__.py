from .base_env import BaseEnv, Error
from .gomoku.game import GameControl
from .connect4.connect4_env import Connect4Env
from .gomoku.gomoku_env import GomokuEnv

Game = BaseEnv  # BASELINE.json calls the env interface "rlzero.games.Game"

__all__ = ['BaseEnv', 'GomokuEnv', 'Connect4Env', 'GameControl', 'Game', 'Error']
