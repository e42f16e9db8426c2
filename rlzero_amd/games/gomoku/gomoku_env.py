"""Host-side Gomoku / TicTacToe environment with the reference's interface.

Mirrors ``GomokuEnv`` of rlzero/games/gomoku/gomoku_env.py (names, arguments, return
values, error behaviour) for the callers of the hot path: ``GameControl`` and
``AlphaZeroPlayer.get_action(game_env)``.  This object is the *single game the caller
steps*; the thousands of positions visited inside the search live as bitboards on the GPU
(csrc/rz_engine.hip).  The rules are kept here as two Python-int bitboards with the same
bit numbering the kernels use (bit = move = h*B + w, gomoku_env.py:227-234), so importing
a position into the engine is a copy of two integers.
"""
from typing import List, Tuple

import numpy as np

from ..base_env import BaseEnv, Error


def _start_masks(size, n):
    """For the 4 line directions: (stride, mask of cells that may START an n-line).
    Same validity conditions as the reference's scan (gomoku_env.py:140-168)."""
    right = down = left = 0
    for h in range(size):
        for w in range(size):
            bit = 1 << (h * size + w)
            if w <= size - n:
                right |= bit
            if h <= size - n:
                down |= bit
            if w >= n - 1:
                left |= bit
    return ((1, right), (size, down), (size + 1, right & down), (size - 1, left & down))


def has_line(stones, masks, n):
    """True iff the bitboard holds n consecutive stones on some row/column/diagonal."""
    for stride, starts in masks:
        run = stones & starts
        for j in range(1, n):
            if not run:
                break
            run &= stones >> (j * stride)
        if run:
            return True
    return False


class GomokuEnv(BaseEnv):
    """Board for the game; ``states`` maps move -> player like the reference."""

    def __init__(self, board_size: int = 8, n_in_row: int = 5, start_player_idx: int = 0) -> None:
        super().__init__()
        self.board_size = board_size
        self.n_in_row = n_in_row
        self.players = [0, 1]
        self.start_player_idx = start_player_idx
        self._current_player = self.players[self.start_player_idx]
        self._leagel_actions = list(range(board_size * board_size))
        self._bits = [0, 0]
        self._masks = None

    # ------------------------------------------------------------------ lifecycle
    def reset(self, start_player_idx: int = 0):
        if self.board_size < self.n_in_row:
            raise Error(f'Board board_size can not less than {self.n_in_row}')
        if start_player_idx not in (0, 1):
            raise Error(f'{start_player_idx} should be 0 (player1 first) or 1 (player2 first)')
        self.start_player_idx = start_player_idx
        self._current_player = self.players[start_player_idx]
        self._leagel_actions = list(range(self.board_size * self.board_size))
        self.states = {}
        self.last_move = -1
        self.info = {}
        self._bits = [0, 0]
        self._masks = _start_masks(self.board_size, self.n_in_row)
        return self.current_state()

    @classmethod
    def from_bitboards(cls, board_size, n_in_row, stones0, stones1, to_move, last_move):
        """Materialise a position held by the engine (for host-side evaluators)."""
        env = cls(board_size, n_in_row)
        env.reset()
        env._bits = [int(stones0), int(stones1)]
        cells = board_size * board_size
        placed = [m for m in range(cells) if ((stones0 | stones1) >> m) & 1]
        if last_move in placed:  # keep dict insertion order ending with the last move
            placed.remove(last_move)
            placed.append(last_move)
        env.states = {m: (0 if (stones0 >> m) & 1 else 1) for m in placed}
        env._leagel_actions = [m for m in range(cells) if not ((stones0 | stones1) >> m) & 1]
        env._current_player = int(to_move)
        env.last_move = int(last_move)
        return env

    def bitboards(self):
        """(stones of player 0, stones of player 1) as Python ints."""
        return self._bits[0], self._bits[1]

    # ------------------------------------------------------------------ stepping
    def step(self, action: int):
        assert action in self._leagel_actions, print(
            f'You input illegal action: {action}, the legal_actions are {self._leagel_actions}.')
        action = int(action)  # numpy integers from np.random.choice: keep the bit math in Python ints
        mover = self._current_player
        self.states[action] = mover
        self._leagel_actions.remove(action)
        self._bits[mover] |= 1 << int(action)
        self.last_move = action
        win, winner = self.has_a_winner()
        reward = 0
        if win:
            reward = 1 if winner == mover else -1
        self._current_player = self.players[1 - self.players.index(mover)]
        return self.current_state(), reward, win, self.info

    def leagel_actions(self):  # (sic) -- the reference's spelling is part of the interface
        return self._leagel_actions

    def legal_actions(self, player):
        return self._leagel_actions

    def current_player(self):
        return self._current_player

    def current_player_index(self):
        return 0 if self._current_player == 1 else 1

    # ------------------------------------------------------------------ rules
    def has_a_winner(self) -> Tuple[bool, int]:
        if self._masks is None:
            self._masks = _start_masks(self.board_size, self.n_in_row)
        for player in (0, 1):
            if has_line(self._bits[player], self._masks, self.n_in_row):
                return True, player
        return False, -1

    def game_end_winner(self):
        win, winner = self.has_a_winner()
        if win:
            return True, winner
        if not len(self._leagel_actions):
            return True, -1
        return False, -1

    def is_terminal(self):
        return self.game_end_winner()[0]

    def get_done_reward(self):
        """Same conditions as the reference (gomoku_env.py:172-194), which still tests the
        player ids 1/2 although ``players`` is [0, 1]."""
        win, winner = self.has_a_winner()
        reward = None
        if winner == 1:
            reward = 1
        elif winner == 2:
            reward = -1
        elif winner == -1 and win:
            reward = 0
        return win, reward

    def returns(self):
        win, winner = self.has_a_winner()
        if winner == 1:
            return [1, -1]
        if winner == 2:
            return [-1, 1]
        return [0, 0]

    def max_utility(self):
        return 1

    # ------------------------------------------------------------------ observation
    def current_state(self) -> np.ndarray:
        """4 x B x B float64 planes from the side to move: own stones, opponent stones,
        last move, colour-to-play (all ones iff an even number of stones)."""
        size = self.board_size
        planes = np.zeros((4, size, size))
        if self.states:
            mine = self._bits[self._current_player]
            theirs = self._bits[1 - self._current_player]
            flat = planes.reshape(4, size * size)
            for m in self.states:
                if (mine >> m) & 1:
                    flat[0, m] = 1.0
                elif (theirs >> m) & 1:
                    flat[1, m] = 1.0
            flat[2, self.last_move] = 1.0
        if len(self.states) % 2 == 0:
            planes[3][:, :] = 1.0
        return planes

    # ------------------------------------------------------------------ coordinates / text
    def move_to_location(self, move: int) -> List:
        return [move // self.board_size, move % self.board_size]

    def location_to_move(self, location: List) -> int:
        if len(location) != 2:
            return -1
        move = location[0] * self.board_size + location[1]
        if move not in range(self.board_size * self.board_size):
            return -1
        return move

    def action_to_string(self, move: int):
        return f'Play row {move // self.board_size + 1}, column {move % self.board_size + 1}'

    def render(self):
        size = self.board_size
        print()
        print(''.join('{0:8}'.format(x) for x in range(size)), end='')
        print('\r\n')
        for i in range(size - 1, -1, -1):
            row = '{0:4d}'.format(i)
            for j in range(size):
                owner = self.states.get(i * size + j, -1)
                row += ('B' if owner == self.players[0] else 'W' if owner == self.players[1] else '_').center(8)
            print(row, end='')
            print('\r\n\r\n')

    def __str__(self):
        return 'Gomoku Board'
