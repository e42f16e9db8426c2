from .connect4_env import Connect4Env

__all__ = ['Connect4Env']
