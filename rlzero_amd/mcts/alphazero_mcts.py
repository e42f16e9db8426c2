"""AlphaZero MCTS search + player with the reference's interface
(rlzero/mcts/alphazero_mcts.py:17-169), searched by the HIP engine.

``AlphaZeroMCTS.simulate`` imports the caller's position into a one-game ``MCTSEngine``
(the caller's env is never modified, the reference deep-copies it per simulation,
alphazero_mcts.py:83), runs ``n_playout`` simulations on the GPU and converts the root
visit counts to probabilities with the reference's numpy expression (:88-92).  The tree
persists on the device between calls, so tree reuse in self-play (``update_with_move``)
behaves as in the reference.

Evaluator plug-in (``policy_value_fn``): a bound ``AlphaZeroAgent.policy_value_fn`` whose
agent lives on a GPU is batched on the device; any other callable is called per leaf on a
materialised ``GomokuEnv`` (exactly the reference's contract, slower).
"""
import os

import numpy as np

from ..engine import (HipNetEvaluator, HostEvaluator, MCTSEngine, NetEvaluator, bits_to_int, int_to_bits)
from .player import Player


def softmax(x):
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


class TreeNodeView(object):
    """Read-only snapshot of one node with the reference's attribute names
    (rlzero/mcts/node.py:17-30)."""

    def __init__(self, snap, slot, occ, parent=None, prior=1.0, visited=True):
        self._snap, self._slot, self._occ, self._parent = snap, slot, occ, parent
        self.explore_count = int(snap['N'][slot]) if visited else 0
        self.total_reward = float(snap['W'][slot]) if visited else 0
        self.prior = prior
        self._visited = visited
        self._kids = None

    @property
    def _children(self):
        if self._kids is None:
            self._kids = {}
            snap = self._snap
            k = int(snap['K'][self._slot]) if self._visited or self._parent is None else 0
            if k > 0:
                fc, nv, pb = int(snap['FC'][self._slot]), int(snap['NV'][self._slot]), int(snap['PB'][self._slot])
                empties = [c for c in range(snap['cells']) if not (self._occ >> c) & 1]
                for r, a in enumerate(empties):
                    seen = r < nv  # only visited children hold a record
                    self._kids[a] = TreeNodeView(snap, fc + r if seen else 0, self._occ | (1 << a), self,
                                                 float(snap['PRI'][pb + r]), visited=seen)
        return self._kids

    children = _children

    @property
    def parent(self):
        return self._parent

    def is_leaf(self):
        return self._children == {}

    def is_root(self):
        return self._parent is None


class AlphaZeroMCTS(object):
    """Monte Carlo tree search guided by a policy-value function."""

    def __init__(self, policy_value_fn, n_playout: int = 1000, c_puct: float = 5,
                 add_noise: bool = False, device=None, score_mode: str = 'uct_ref', sims_in_flight: int = 1) -> None:
        self.score_mode = score_mode  # 'uct_ref' = the reference's rule (parity); 'puct' = opt-in
        # opt-in, NOT the reference's algorithm: K > 1 simulations of the tree share one evaluator batch (virtual loss)
        self.sims_in_flight = max(1, int(sims_in_flight))
        self.policy_value_fn = policy_value_fn
        self.n_playout = n_playout
        self._c_puct = c_puct
        self.add_noise = add_noise  # Dirichlet noise on the stored priors (never read by 'uct_ref')
        self._device = device
        self._engine = None
        self._evaluator = None
        self._root_occ = 0
        self._legal = ()

    # ------------------------------------------------------------------ engine binding
    def _bind(self, game_env):
        size, n_row = game_env.board_size, game_env.n_in_row
        kind = getattr(game_env, 'game_kind', 'gomoku')
        eng = self._engine
        if eng is not None and (eng.game, eng.board_size, eng.n_in_row) == (kind, size, n_row) and \
                eng.n_playout >= self.n_playout and eng.c_puct == float(self._c_puct) and \
                getattr(self, '_bound_k', 1) == self.sims_in_flight:
            return eng
        if eng is not None:
            eng.close()
        agent = getattr(self.policy_value_fn, '__self__', None)
        net = getattr(agent, 'policy_value_net', None)
        agent_dev = str(getattr(agent, 'device', 'cpu'))
        rows, cols = (size, size) if kind == 'gomoku' else size
        fast = net is not None and (getattr(agent, 'board_size', None), getattr(agent, 'board_width', rows)) == \
            (rows, cols) and \
            agent_dev.startswith('cuda') and \
            getattr(self.policy_value_fn, '__func__', None) is getattr(type(agent), 'policy_value_fn', None)
        device = self._device or (agent_dev if fast else os.environ.get('RLZERO_DEVICE', 'cuda:0'))
        eng = MCTSEngine(size, n_row, n_games=1, n_playout=self.n_playout, c_puct=self._c_puct,
                         device=device, score_mode=self.score_mode, add_noise=self.add_noise, game=kind,
                         sims_in_flight=self.sims_in_flight if fast else 1,
                         noise_seed=int(np.random.randint(0, 2 ** 31 - 1)) if self.add_noise else 0)
        if fast:
            from ..games.gomoku.policy_value_net import PolicyValueNet
            # the reference architecture runs on the hand-written fused kernels (csrc/rz_net.hip);
            # any other nn.Module with the same call signature goes through PyTorch-ROCm
            self._evaluator = HipNetEvaluator(net, (rows, cols, eng.n_actions), eng.device, max_boards=eng.n_leaves) \
                if type(net) is PolicyValueNet else NetEvaluator(net)
        else:
            from ..games.connect4.connect4_env import Connect4Env
            from ..games.gomoku.gomoku_env import GomokuEnv
            if kind == 'connect4':
                make = lambda s0, s1, to_move, last: Connect4Env.from_bitboards(rows, cols, n_row, s0, s1, to_move, last)  # noqa: E731
            else:
                make = lambda s0, s1, to_move, last: GomokuEnv.from_bitboards(size, n_row, s0, s1, to_move, last)  # noqa: E731
            self._evaluator = HostEvaluator(lambda env: self.policy_value_fn(env), make)
        self._engine = eng
        self._bound_k = self.sims_in_flight
        # device evaluators: the simulation loop is replayed from a hipGraph of 8 simulations (three launches each) --
        # one game per launch is bound by launch latency, not by the kernels.  Captured now, on the still empty tree.
        self._graph_sims = 0
        if not isinstance(self._evaluator, HostEvaluator) and os.environ.get('RLZERO_NO_GRAPH') != '1':
            eng.reset_games()
            self._graph_sims = eng.graph_chunk(8)
            try:
                eng.warm_graph(self._evaluator, self._graph_sims)
            except RuntimeError:
                # an arbitrary nn.Module (NetEvaluator) may do something a stream capture does not allow: launch it
                # kernel by kernel instead (the hand-written evaluator always captures)
                if isinstance(self._evaluator, HipNetEvaluator):
                    raise
                eng.torch.cuda.synchronize(eng.device)
                self._graph_sims = 0
            eng.reset_games()
        return eng

    def _import_root(self, game_env):
        eng = self._bind(game_env)
        s0, s1 = _bitboards_of(game_env)
        stones = np.array([[int_to_bits(s0), int_to_bits(s1)]], dtype=np.uint64)
        eng.set_roots(stones, [game_env.current_player()], [getattr(game_env, 'last_cell', game_env.last_move)])
        self._root_occ = s0 | s1
        self._legal = tuple(game_env.leagel_actions())
        return eng

    # ------------------------------------------------------------------ reference API
    def simulate(self, game_env, temperature: float = 1e-3):
        """All ``n_playout`` simulations, then (actions, probabilities) of the root
        children: softmax(1/T * log(N + 1e-10)) (alphazero_mcts.py:73-94)."""
        eng = self._import_root(game_env)
        if isinstance(self._evaluator, HipNetEvaluator):
            self._evaluator.refresh_if_changed()  # the learner may have stepped since the last move
        eng.simulate(self._evaluator, self.n_playout, use_graph=self._graph_sims > 0, sims_per_graph=max(self._graph_sims, 1))
        visits = eng.root_visits()[0]
        eng.poll_errors()   # (8 bytes; the full statistics only when a flag is set)
        if isinstance(self._evaluator, HipNetEvaluator) and self._evaluator.needs_obs:
            self._evaluator.hip.check_flags()  # (only float planes can leave the f16 range: positions are 0 / 1 planes by construction)
        acts = self._legal
        counts = np.array([int(visits[a]) for a in acts])
        act_probs = softmax(1.0 / temperature * np.log(counts + 1e-10))
        return acts, act_probs

    def update_with_move(self, last_move):
        """Keep the subtree below ``last_move`` or start a fresh tree (:96-103)."""
        if self._engine is None:
            return
        move = int(last_move)
        if move not in self._legal:  # e.g. -1: the reference falls back to a new root
            move = -1
        self._engine.advance([move])
        self._legal = ()

    @property
    def _root(self):
        if self._engine is None:
            return TreeNodeView({'N': [0], 'W': [0], 'FC': [-1], 'NV': [0], 'K': [0], 'PB': [-1], 'PRI': [],
                                 'cells': 0}, 0, 0)
        snap = self._engine.arena(0)
        snap['cells'] = self._engine.n_cells
        stones, _, _ = self._engine.get_roots()
        occ = bits_to_int(stones[0, 0]) | bits_to_int(stones[0, 1])
        return TreeNodeView(snap, 0, occ, None, snap['root_prior'])

    def __str__(self):
        return 'AlphaZeroMCTS'


def _bitboards_of(game_env):
    if hasattr(game_env, 'bitboards'):
        return game_env.bitboards()
    s = [0, 0]
    for move, player in game_env.states.items():  # any env with the reference's fields
        s[0 if player == game_env.players[0] else 1] |= 1 << int(move)
    return s[0], s[1]


class AlphaZeroPlayer(Player):
    """AI player based on MCTS (alphazero_mcts.py:109-169)."""

    def __init__(self, policy_value_fn, n_playout: int = 1000, c_puct: float = 5,
                 is_selfplay: bool = False, player_id: int = 0, player_name: str = '',
                 device=None, score_mode: str = 'uct_ref', sims_in_flight: int = 1) -> None:
        super().__init__(player_id, player_name)
        self.is_selfplay = is_selfplay
        self.add_noise = is_selfplay
        self.mcts = AlphaZeroMCTS(policy_value_fn, n_playout=n_playout, c_puct=c_puct,
                                  add_noise=self.add_noise, device=device, score_mode=score_mode,
                                  sims_in_flight=sims_in_flight)

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def get_action(self, game_env, temperature: float = 1e-3, return_prob: bool = False):
        sensible_moves = game_env.leagel_actions()
        move_probs = np.zeros(getattr(game_env, 'n_actions', None) or game_env.board_size * game_env.board_size)
        if len(sensible_moves) == 0:
            print('WARNING: the board is full')
            return None
        acts, probs = self.mcts.simulate(game_env, temperature)
        move_probs[list(acts)] = probs
        move = np.random.choice(acts, p=probs)
        if self.is_selfplay:
            self.mcts.update_with_move(move)  # tree reuse
        else:
            move = np.random.choice(acts, p=probs)  # the reference draws twice (:157)
            self.mcts.update_with_move(-1)
        return (move, move_probs) if return_prob else move

    def __str__(self):
        return 'AlphaZeroPlayer, id: {}, name: {}.'.format(self.get_player_id(), self.get_player_name())
