"""Oracle: Gomoku / TicTacToe rules (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``rlzero/games/gomoku/gomoku_env.py`` of the reference:

* board / legal list / player flip ............ gomoku_env.py:33-47, 49-70, 72
* n-in-row scan ............................... gomoku_env.py:116-170
* end-of-game rule ............................ gomoku_env.py:196-203
* 4-plane observation ......................... gomoku_env.py:95-114, 227-234

The board is a flat list of cell owners (-1 empty, 0 / 1 the two players); the
reference keeps a ``{move: player}`` dict and an ascending legal list, both are
derived views here (``states`` / ``leagel_actions``) so evaluators written
against the reference's env interface work unchanged on this object.
"""
import numpy as np


class RefGomoku(object):

    players = (0, 1)

    def __init__(self, board_size=8, n_in_row=5):
        self.board_size = int(board_size)
        self.n_in_row = int(n_in_row)
        self.reset()

    # -- gomoku_env.py:33-47 -------------------------------------------------
    def reset(self, start_player_idx=0):
        if self.board_size < self.n_in_row:
            raise ValueError('board_size smaller than n_in_row')
        if start_player_idx not in (0, 1):
            raise ValueError('start_player_idx must be 0 or 1')
        size = self.board_size * self.board_size
        self.cells = [-1] * size
        self.order = []  # moves in the order played (dict insertion order)
        self.to_move = self.players[start_player_idx]
        self.last_move = -1
        return self.current_state()

    def clone(self):
        other = object.__new__(RefGomoku)
        other.board_size = self.board_size
        other.n_in_row = self.n_in_row
        other.cells = list(self.cells)
        other.order = list(self.order)
        other.to_move = self.to_move
        other.last_move = self.last_move
        return other

    # -- views with the reference's names ------------------------------------
    @property
    def states(self):
        return {m: self.cells[m] for m in self.order}

    def leagel_actions(self):  # (sic) gomoku_env.py:72 -- ascending
        return [m for m, owner in enumerate(self.cells) if owner < 0]

    def current_player(self):  # gomoku_env.py:271-272
        return self.to_move

    # -- gomoku_env.py:49-70 ---------------------------------------------------
    def step(self, action):
        action = int(action)
        assert 0 <= action < len(self.cells) and self.cells[action] < 0, \
            'illegal action %r' % (action, )
        mover = self.to_move
        self.cells[action] = mover
        self.order.append(action)
        self.last_move = action
        won, winner = self.has_a_winner()
        reward = 0
        if won:
            reward = 1 if winner == mover else -1
        self.to_move = 1 - mover
        return reward, won

    # -- gomoku_env.py:116-170 -------------------------------------------------
    def has_a_winner(self):
        size, n, cells = self.board_size, self.n_in_row, self.cells
        if len(self.order) < 2 * n - 1:  # :133 (never changes the answer)
            return False, -1
        for m, owner in enumerate(cells):
            if owner < 0:
                continue
            h, w = divmod(m, size)
            room_right = w <= size - n
            room_down = h <= size - n
            room_left = w >= n - 1
            # (condition, stride) for: row, column, diagonal, anti-diagonal
            for ok, stride in ((room_right, 1), (room_down, size),
                               (room_right and room_down, size + 1),
                               (room_left and room_down, size - 1)):
                if not ok:
                    continue
                j = 1
                while j < n and cells[m + j * stride] == owner:
                    j += 1
                if j == n:
                    return True, owner
        return False, -1

    # -- gomoku_env.py:196-203 -------------------------------------------------
    def game_end_winner(self):
        won, winner = self.has_a_winner()
        if won:
            return True, winner
        if len(self.order) == len(self.cells):
            return True, -1
        return False, -1

    # -- gomoku_env.py:95-114 --------------------------------------------------
    def current_state(self):
        size = self.board_size
        planes = np.zeros((4, size, size))
        if self.order:
            for m in self.order:
                h, w = divmod(m, size)  # gomoku_env.py:227-234, no flip
                planes[0 if self.cells[m] == self.to_move else 1, h, w] = 1.0
            h, w = divmod(self.last_move, size)
            planes[2, h, w] = 1.0
        if len(self.order) % 2 == 0:
            planes[3, :, :] = 1.0
        return planes

    # helpers used by tests -----------------------------------------------------
    def bitboards(self):
        """(stones of player 0, stones of player 1) as Python ints, bit = move."""
        b = [0, 0]
        for m in self.order:
            b[self.cells[m]] |= 1 << m
        return b[0], b[1]

    @classmethod
    def from_moves(cls, board_size, n_in_row, moves, start_player_idx=0):
        env = cls(board_size, n_in_row)
        env.reset(start_player_idx)
        for m in moves:
            env.step(m)
        return env
