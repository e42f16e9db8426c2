from .alphazero_mcts import AlphaZeroMCTS, AlphaZeroPlayer
from .player import HumanPlayer, Player

MCTSPlayer = AlphaZeroPlayer  # the name BASELINE.json uses for the reference's AlphaZeroPlayer

__all__ = ['AlphaZeroMCTS', 'AlphaZeroPlayer', 'MCTSPlayer', 'Player', 'HumanPlayer']
