"""Oracle: Connect4 rules (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference implements no Connect4 (docs/open-spiel_alphazero.md:58 only mentions
OpenSpiel's): PARITY UNPINNED against the reference.  This is the naive twin that pins the
build-defined rules of rlzero_amd/games/connect4 and of the RZ_GAME_CONNECT4 kernels:
rows x cols grid (6 x 7), an action is a column, the stone drops to the lowest empty cell
(cell = row*cols + col, row 0 at the bottom), first n_in_row (4) in a row / column / diagonal
wins -- scanned cell by cell like the reference's Gomoku scan (gomoku_env.py:136-168) -- full
board without a line is a tie; observation = the Gomoku 4 planes (gomoku_env.py:95-114).
It exposes the env interface the oracle search (oracle/mcts_ref.py) expects.
"""
import numpy as np


class RefConnect4(object):
    players = (0, 1)
    game_kind = 'connect4'

    def __init__(self, rows=6, cols=7, n_in_row=4):
        self.rows, self.cols, self.n_in_row = rows, cols, n_in_row
        self.board_size = (rows, cols)
        self.n_actions = cols
        self.reset()

    def reset(self, start_player_idx=0):
        self.cells = [-1] * (self.rows * self.cols)
        self.order = []
        self.to_move = self.players[start_player_idx]
        self.last_move = -1
        self.last_cell = -1
        return self.current_state()

    def clone(self):
        other = object.__new__(RefConnect4)
        other.__dict__.update(self.__dict__)
        other.cells = list(self.cells)
        other.order = list(self.order)
        return other

    @property
    def states(self):
        return {c: self.cells[c] for c in self.order}

    def leagel_actions(self):
        top = (self.rows - 1) * self.cols
        return [c for c in range(self.cols) if self.cells[top + c] < 0]

    def current_player(self):
        return self.to_move

    def step(self, action):
        action = int(action)
        row = 0
        while row < self.rows and self.cells[row * self.cols + action] >= 0:
            row += 1
        assert 0 <= action < self.cols and row < self.rows, 'illegal action %r' % (action, )
        cell = row * self.cols + action
        self.cells[cell] = self.to_move
        self.order.append(cell)
        self.last_move, self.last_cell = action, cell
        self.to_move = 1 - self.to_move

    def has_a_winner(self):
        rows, cols, n, cells = self.rows, self.cols, self.n_in_row, self.cells
        for m, owner in enumerate(cells):
            if owner < 0:
                continue
            h, w = divmod(m, cols)
            right, down, left = w <= cols - n, h <= rows - n, w >= n - 1
            for ok, stride in ((right, 1), (down, cols), (right and down, cols + 1), (left and down, cols - 1)):
                if ok and all(cells[m + j * stride] == owner for j in range(1, n)):
                    return True, owner
        return False, -1

    def game_end_winner(self):
        won, winner = self.has_a_winner()
        if won:
            return True, winner
        if len(self.order) == len(self.cells):
            return True, -1
        return False, -1

    def current_state(self):
        planes = np.zeros((4, self.rows, self.cols))
        flat = planes.reshape(4, -1)
        if self.order:
            for c in self.order:
                flat[0 if self.cells[c] == self.to_move else 1, c] = 1.0
            flat[2, self.last_cell] = 1.0
        if len(self.order) % 2 == 0:
            planes[3, :, :] = 1.0
        return planes

    def bitboards(self):
        b = [0, 0]
        for c in self.order:
            b[self.cells[c]] |= 1 << c
        return b[0], b[1]

    @classmethod
    def from_moves(cls, moves, rows=6, cols=7, n_in_row=4):
        env = cls(rows, cols, n_in_row)
        for m in moves:
            env.step(m)
        return env
