"""Game loops with the reference's interface (rlzero/games/gomoku/game.py:12-137).

``GameControl`` drives ONE game through ``player.get_action`` exactly like the reference
(same return tuples, same z rule, same ``reset_player()`` at the end of self-play); the
many-games-in-lock-step counterpart that feeds the GPU is ``rlzero_amd.selfplay``.
"""
import numpy as np

from ..base_env import Error
from .gomoku_env import GomokuEnv


class GameControl(object):
    """game server."""

    def __init__(self, game_env: GomokuEnv) -> None:
        self.game_env = game_env
        self.visualTool = None

    def set_player_symbol(self, start_player) -> None:
        first = self.game_env.players[start_player] == self.game_env.players[0]
        self.player1_symbol, self.player2_symbol = ('X', 'O') if first else ('O', 'X')

    def graphic(self, game_env, player1, player2):
        """ASCII board (game.py:28-59)."""
        size = game_env.board_size
        id1 = player1 if isinstance(player1, int) else player1.get_player_id()
        id2 = player2 if isinstance(player2, int) else player2.get_player_id()
        print('Player', player1, self.player1_symbol.rjust(3))
        print('Player', player2, self.player2_symbol.rjust(3))
        print()
        print(''.join('{0:8}'.format(x) for x in range(size)), end='')
        print('\r\n')
        for i in range(size - 1, -1, -1):
            row = '{0:4d}'.format(i)
            for j in range(size):
                owner = game_env.states.get(i * size + j, -1)
                row += (self.player1_symbol if owner == id1 else
                        self.player2_symbol if owner == id2 else '_').center(8)
            print(row, end='')
            print('\r\n\r\n')

    def start_play(self, player1, player2, start_player: int = 0, is_shown: bool = True) -> int:
        """Two players alternate until the game ends; returns the winner id or -1
        (game.py:61-94; like the reference, ``reset()`` ignores ``start_player``)."""
        if start_player not in (0, 1):
            raise Error(f'{start_player} should be 0 (player1 first) or 1 (player2 first)')
        env = self.game_env
        env.reset()
        p1, p2 = env.players
        player1.set_player_id(p1)
        player2.set_player_id(p2)
        self.set_player_symbol(start_player)
        seats = {p1: player1, p2: player2}
        if is_shown:
            self.graphic(env, player1, player2)
        while True:
            move = seats[env.current_player()].get_action(env)
            env.step(move)
            if is_shown:
                self.graphic(env, player1, player2)
            end, winner = env.game_end_winner()
            if end:
                if is_shown:
                    print('Game end. Winner is', seats[winner]) if winner != -1 else print('Game end. Tie')
                return winner

    def start_self_play(self, player, is_shown: bool = False, temperature: float = 1e-3):
        """One self-play game -> (winner, zip(states, mcts_probs, winners_z))
        (game.py:96-134)."""
        env = self.game_env
        env.reset()
        p1, p2 = env.players
        states, mcts_probs, movers = [], [], []
        self.set_player_symbol(start_player=0)
        while True:
            move, move_probs = player.get_action(env, temperature=temperature, return_prob=True)
            states.append(env.current_state())
            mcts_probs.append(move_probs)
            movers.append(env.current_player())
            env.step(move)
            if is_shown:
                self.graphic(env, p1, p2)
            end, winner = env.game_end_winner()
            if end:
                winners_z = np.zeros(len(movers))
                if winner != -1:
                    seat = np.array(movers)
                    winners_z[seat == winner] = 1.0
                    winners_z[seat != winner] = -1.0
                player.reset_player()
                if is_shown:
                    print('Game end. Winner is player:', winner) if winner != -1 else print('Game end. Tie')
                return winner, zip(states, mcts_probs, winners_z)

    def __str__(self):
        return 'Game'
