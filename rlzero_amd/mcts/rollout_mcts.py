"""Pure-MCTS opponent with the reference's interface (rlzero/mcts/rollout_mcts.py:10-140):
same tree and UCT selection as the AlphaZero search, uniform priors, leaf value from a random
play-out, most visited move, tree reset after every move.  Used by the reference's trainer as
the evaluation opponent (tools/train_alphazero.py:139-163).

The tree lives in the HIP engine; the play-outs run on the device on bitboards
(csrc/rz_engine.hip, k_eval_rollout).  The reference draws its play-out moves from numpy's
global stream (``np.random.rand(k)`` per ply, :99); here every ``get_action`` draws ONE integer
from that stream as the seed of a counter-based generator, so ``np.random.seed`` still makes a
game reproducible, though not move-for-move equal to the reference's stream.
"""
import os

import numpy as np

from ..engine import MCTSEngine, RolloutEvaluator, int_to_bits
from .alphazero_mcts import TreeNodeView, _bitboards_of
from .player import Player

_M64 = (1 << 64) - 1


def _mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def rollout_pick(seed, game, sim, ply, k):
    """Index (0..k-1) of the legal move the device play-out takes at ``ply`` of simulation
    ``sim`` of ``game`` (host twin of k_eval_rollout's generator, for parity tests)."""
    key = _mix64(_mix64((seed ^ game) & _M64) ^ sim)
    u = _mix64(key ^ ply) >> 32
    return (u * k) >> 32


class RolloutMCTS(object):

    def __init__(self, n_playout: int = 1000, c_puct: float = 5.0, n_limit: int = 1000, device=None) -> None:
        self.n_playout = n_playout
        self._c_puct = c_puct
        self.n_limit = n_limit
        self._device = device
        self._engine = None
        self._legal = ()
        self.seed = None  # fixed seed for tests; None = one draw from numpy's global stream per move

    def _bind(self, game_env):
        size, n_row = game_env.board_size, game_env.n_in_row
        kind = getattr(game_env, 'game_kind', 'gomoku')
        eng = self._engine
        if eng is not None and (eng.game, eng.board_size, eng.n_in_row) == (kind, size, n_row) and \
                eng.n_playout >= self.n_playout and eng.c_puct == float(self._c_puct):
            return eng
        if eng is not None:
            eng.close()
        self._engine = MCTSEngine(size, n_row, n_games=1, n_playout=self.n_playout, c_puct=self._c_puct,
                                  device=self._device or os.environ.get('RLZERO_DEVICE', 'cuda:0'), game=kind)
        return self._engine

    def simulate(self, game_env, temperature: float = 0.001):
        """n_playout play-outs; returns the most visited root move (first maximum)."""
        eng = self._bind(game_env)
        s0, s1 = _bitboards_of(game_env)
        eng.set_roots(np.array([[int_to_bits(s0), int_to_bits(s1)]], dtype=np.uint64),
                      [game_env.current_player()], [getattr(game_env, 'last_cell', game_env.last_move)])
        self._legal = tuple(game_env.leagel_actions())
        seed = self.seed if self.seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        self.evaluator = RolloutEvaluator(seed, self.n_limit)
        eng.simulate(self.evaluator, self.n_playout)
        visits = eng.root_visits()[0]
        eng.check()
        counts = [int(visits[a]) for a in self._legal]
        return self._legal[counts.index(max(counts))]

    def update_with_move(self, last_move: int):
        if self._engine is None:
            return
        move = int(last_move)
        self._engine.advance([move if move in self._legal else -1])
        self._legal = ()

    # host-side policies kept for interface compatibility (rollout_mcts.py:96-108)
    def rollout_policy(self, game_env):
        legal = game_env.leagel_actions()
        return zip(legal, np.random.rand(len(legal)))

    def policy_value_fn(self, game_env):
        legal = game_env.leagel_actions()
        return zip(legal, np.ones(len(legal)) / len(legal))

    @property
    def _root(self):
        if self._engine is None:
            return TreeNodeView({'N': [0], 'W': [0], 'FC': [-1], 'NV': [0], 'K': [0], 'PB': [-1], 'PRI': [],
                                 'cells': 0}, 0, 0)
        from ..engine import bits_to_int
        snap = self._engine.arena(0)
        snap['cells'] = self._engine.n_cells
        stones, _, _ = self._engine.get_roots()
        return TreeNodeView(snap, 0, bits_to_int(stones[0, 0]) | bits_to_int(stones[0, 1]), None, 1.0)

    def __str__(self):
        return 'RolloutMCTS'


class RolloutPlayer(Player):

    def __init__(self, n_playout: int = 1000, c_puct: float = 5, player_id: int = 0,
                 player_name: str = '', device=None) -> None:
        super().__init__(player_id, player_name)
        self.mcts = RolloutMCTS(n_playout, c_puct, device=device)

    def reset_player(self) -> None:
        self.mcts.update_with_move(-1)

    def get_action(self, game_env, **kwargs):
        if len(game_env.leagel_actions()) == 0:
            print('WARNING: the board is full')
            return None
        move = self.mcts.simulate(game_env)
        self.mcts.update_with_move(-1)
        return move

    def __str__(self):
        return 'RolloutPlayer, id: {}, name: {}.'.format(self.get_player_id(), self.get_player_name())
