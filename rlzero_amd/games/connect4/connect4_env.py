"""Connect4 with the interface of the reference's ``GomokuEnv`` (BASELINE.json config 3).

The reference has no Connect4 (docs/open-spiel_alphazero.md:58 only names OpenSpiel's), so the
rules are build-defined and pinned by a naive cell-by-cell twin that only the tests use:
``board_height x board_width`` (6 x 7), ``n_in_row`` 4, an ACTION is a column, the stone drops to
the lowest empty cell, cells are numbered ``row * board_width + column`` with row 0 at the
bottom, players [0, 1], player 0 first.  Everything else follows the Gomoku conventions so the
search, the players and ``GameControl`` work unchanged: ``states`` maps cell -> player,
``leagel_actions()`` is the ascending list of playable columns, ``current_state()`` is the same
4 planes (own stones, opponent stones, last stone, colour to play) as [4, rows, cols].
"""
import numpy as np

from ..base_env import BaseEnv, Error
from ..gomoku.gomoku_env import has_line


def _start_masks(rows, cols, n):
    right = down = left = 0
    for h in range(rows):
        for w in range(cols):
            bit = 1 << (h * cols + w)
            if w <= cols - n:
                right |= bit
            if h <= rows - n:
                down |= bit
            if w >= n - 1:
                left |= bit
    return ((1, right), (cols, down), (cols + 1, right & down), (cols - 1, left & down))


class Connect4Env(BaseEnv):
    game_kind = 'connect4'

    def __init__(self, board_height: int = 6, board_width: int = 7, n_in_row: int = 4) -> None:
        super().__init__()
        self.board_height, self.board_width, self.n_in_row = board_height, board_width, n_in_row
        self.board_size = (board_height, board_width)
        self.n_actions = board_width
        self.players = [0, 1]
        self._masks = _start_masks(board_height, board_width, n_in_row)
        self.reset()

    def reset(self, start_player_idx: int = 0):
        if start_player_idx not in (0, 1):
            raise Error(f'{start_player_idx} should be 0 (player1 first) or 1 (player2 first)')
        if max(self.board_height, self.board_width) < self.n_in_row:
            raise Error(f'Board can not be smaller than {self.n_in_row}')
        self._current_player = self.players[start_player_idx]
        self.states = {}
        self.heights = [0] * self.board_width
        self.last_move = -1   # last action (column)
        self.last_cell = -1   # the cell it occupied
        self.info = {}
        self._bits = [0, 0]
        return self.current_state()

    @classmethod
    def from_bitboards(cls, board_height, board_width, n_in_row, stones0, stones1, to_move, last_cell):
        env = cls(board_height, board_width, n_in_row)
        env._bits = [int(stones0), int(stones1)]
        occ = env._bits[0] | env._bits[1]
        cells = [c for c in range(board_height * board_width) if (occ >> c) & 1]
        if last_cell in cells:
            cells.remove(last_cell)
            cells.append(last_cell)
        env.states = {c: (0 if (stones0 >> c) & 1 else 1) for c in cells}
        env.heights = [sum((occ >> (r * board_width + c)) & 1 for r in range(board_height))
                       for c in range(board_width)]
        env._current_player = int(to_move)
        env.last_cell = int(last_cell)
        env.last_move = int(last_cell) % board_width if last_cell >= 0 else -1
        return env

    def bitboards(self):
        return self._bits[0], self._bits[1]

    def leagel_actions(self):
        return [c for c in range(self.board_width) if self.heights[c] < self.board_height]

    def legal_actions(self, player):
        return self.leagel_actions()

    def current_player(self):
        return self._current_player

    def step(self, action: int):
        action = int(action)
        assert 0 <= action < self.board_width and self.heights[action] < self.board_height, \
            f'You input illegal action: {action}, the legal_actions are {self.leagel_actions()}.'
        mover = self._current_player
        cell = self.heights[action] * self.board_width + action
        self.heights[action] += 1
        self.states[cell] = mover
        self._bits[mover] |= 1 << cell
        self.last_move, self.last_cell = action, cell
        win, winner = self.has_a_winner()
        reward = 0
        if win:
            reward = 1 if winner == mover else -1
        self._current_player = 1 - mover
        return self.current_state(), reward, win, self.info

    def has_a_winner(self):
        for player in (0, 1):
            if has_line(self._bits[player], self._masks, self.n_in_row):
                return True, player
        return False, -1

    def game_end_winner(self):
        win, winner = self.has_a_winner()
        if win:
            return True, winner
        if len(self.states) == self.board_height * self.board_width:
            return True, -1
        return False, -1

    def is_terminal(self):
        return self.game_end_winner()[0]

    def returns(self):
        win, winner = self.has_a_winner()
        return [0, 0] if not win else ([1, -1] if winner == 0 else [-1, 1])

    def current_state(self) -> np.ndarray:
        rows, cols = self.board_height, self.board_width
        planes = np.zeros((4, rows, cols))
        flat = planes.reshape(4, rows * cols)
        if self.states:
            for cell, owner in self.states.items():
                flat[0 if owner == self._current_player else 1, cell] = 1.0
            flat[2, self.last_cell] = 1.0
        if len(self.states) % 2 == 0:
            planes[3][:, :] = 1.0
        return planes

    def render(self):
        for r in range(self.board_height - 1, -1, -1):
            print(' '.join('XO'[self.states[r * self.board_width + c]] if r * self.board_width + c in self.states
                           else '.' for c in range(self.board_width)))
        print(' '.join(str(c) for c in range(self.board_width)))

    def __str__(self):
        return 'Connect4 Board'
