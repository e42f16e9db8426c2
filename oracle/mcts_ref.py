"""Oracle: AlphaZero MCTS search, player and game loop (TEST INFRASTRUCTURE).

Restates, one simulation at a time and in IEEE fp64 exactly as CPython does it:

* tree node + UCT select + expand + backup ... rlzero/mcts/node.py:17-30, 32-42,
                                                44-73, 75-88, 119-144
* playout / simulate / tree reuse ............ rlzero/mcts/alphazero_mcts.py:42-71,
                                                73-94, 96-103
* player (move sampling, reuse vs reset) ..... rlzero/mcts/alphazero_mcts.py:109-165
* game loops .................................. rlzero/games/gomoku/game.py:61-94, 96-134

Deliberate, documented difference: Dirichlet noise (node.py:63-69) only perturbs the
stored prior, which the reference's selection rule (UCT, node.py:41-42,75-88) never
reads, so it is not drawn here; the move is drawn through an injectable
``choice(acts, probs)`` callable (default ``numpy.random.choice``, exactly the
reference's call at alphazero_mcts.py:148).
"""
import math

import numpy as np

INF = float('inf')


class RefNode(object):
    """node.py:17-30.  Children are two parallel lists in ascending action order
    (the reference's dict insertion order, node.py:62-73)."""
    __slots__ = ('parent', 'acts', 'kids', 'n', 'w', 'p')

    def __init__(self, parent=None, prior=1.0):
        self.parent = parent
        self.acts = []
        self.kids = []
        self.n = 0  # explore_count
        self.w = 0  # total_reward (int 0 until the first float is added)
        self.p = prior

    def child(self, action):
        return self.kids[self.acts.index(action)]


def uct_score(parent_n, child_n, child_w, c_puct):
    """node.py:75-88 -- two separately rounded ops in q + c*u, no FMA."""
    if parent_n == 0 or child_n == 0:
        return INF
    q = child_w / child_n
    u = math.sqrt(math.log(parent_n) / child_n)
    return q + c_puct * u


def puct_score(parent_n, child_n, child_w, child_p, c_puct):
    """node.py:105-117 (dead code in the reference) with Q = 0 at N = 0 instead of its division
    by zero: exploration_score + c_puct * (prior * sqrt(parent N) / (N + 1)).  The opt-in
    RZ_SCORE_PUCT mode of the engine; parity for it is pinned by this restatement only."""
    q = child_w / child_n if child_n > 0 else 0.0
    u = float(child_p) * math.sqrt(parent_n) / (child_n + 1)
    return q + c_puct * u


def select_child(node, c_puct, score_mode='uct_ref'):
    """node.py:32-42 -- Python max(): the first maximal child wins."""
    if not node.kids:
        raise ValueError('Node has no children.')
    best_i, best_s = 0, None
    for i, kid in enumerate(node.kids):
        s = uct_score(node.n, kid.n, kid.w, c_puct) if score_mode == 'uct_ref' else \
            puct_score(node.n, kid.n, kid.w, kid.p, c_puct)
        if best_s is None or s > best_s:
            best_i, best_s = i, s
    return node.acts[best_i], node.kids[best_i]


def expand(node, action_priors):
    """node.py:44-73 without the (never read) Dirichlet perturbation."""
    for action, prob in action_priors:
        if action not in node.acts:
            node.acts.append(action)
            node.kids.append(RefNode(node, prob))


def backup(leaf, value):
    """node.py:119-144: leaf gets ``value``, its parent ``-value``, ... to the root."""
    node = leaf
    while node is not None:
        node.n += 1
        node.w += value
        value = -value
        node = node.parent


def softmax(x):
    """alphazero_mcts.py:10-14."""
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


class RefSearch(object):
    """alphazero_mcts.py:17-103."""

    def __init__(self, policy_value_fn, n_playout=1000, c_puct=5, score_mode='uct_ref'):
        self.score_mode = score_mode
        self.root = RefNode(None, 1.0)
        self.policy_value_fn = policy_value_fn
        self.n_playout = n_playout
        self.c_puct = c_puct
        self.leaf_log = None  # optional list: (path actions, value fed to backup)

    def playout(self, env):
        """alphazero_mcts.py:42-71; ``env`` is a private copy and is modified."""
        node = self.root
        path = []
        while node.kids:
            action, node = select_child(node, self.c_puct, self.score_mode)
            env.step(action)
            path.append(action)
        action_priors, leaf_value = self.policy_value_fn(env)  # always called (:59)
        ended, winner = env.game_end_winner()
        if not ended:
            expand(node, action_priors)
        elif winner == -1:
            leaf_value = 0.0
        else:
            leaf_value = 1.0 if winner == env.current_player() else -1.0
        if self.leaf_log is not None:
            self.leaf_log.append((tuple(path), float(leaf_value)))
        backup(node, -leaf_value)

    def simulate(self, env, temperature=1e-3):
        """alphazero_mcts.py:73-94."""
        for _ in range(self.n_playout):
            self.playout(env.clone())
        acts = tuple(self.root.acts)
        visits = [kid.n for kid in self.root.kids]
        probs = softmax(1.0 / temperature * np.log(np.array(visits) + 1e-10))
        return acts, probs

    def update_with_move(self, last_move):
        """alphazero_mcts.py:96-103."""
        if last_move in self.root.acts:
            self.root = self.root.child(last_move)
            self.root.parent = None
        else:
            self.root = RefNode(None, 1.0)


class RefPlayer(object):
    """alphazero_mcts.py:109-165."""

    def __init__(self, policy_value_fn, n_playout=1000, c_puct=5, is_selfplay=False,
                 choice=None):
        self.is_selfplay = is_selfplay
        self.mcts = RefSearch(policy_value_fn, n_playout, c_puct)
        self.choice = choice if choice is not None else \
            (lambda acts, probs: np.random.choice(acts, p=probs))
        self.player_id = 0

    def set_player_id(self, player_id):
        self.player_id = player_id

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def get_action(self, env, temperature=1e-3, return_prob=False):
        size = getattr(env, 'n_actions', None) or env.board_size * env.board_size
        move_probs = np.zeros(size)
        if not env.leagel_actions():
            print('WARNING: the board is full')
            return None
        acts, probs = self.mcts.simulate(env, temperature)
        move_probs[list(acts)] = probs
        move = self.choice(acts, probs)
        if self.is_selfplay:
            self.mcts.update_with_move(move)
        else:
            move = self.choice(acts, probs)  # second draw is the one used (:157)
            self.mcts.update_with_move(-1)
        return (move, move_probs) if return_prob else move


def self_play_game(env, player, temperature=1e-3):
    """game.py:96-134 -> (winner, [(state, pi, z), ...], moves)."""
    env.reset()
    states, pis, movers, moves = [], [], [], []
    while True:
        move, pi = player.get_action(env, temperature=temperature, return_prob=True)
        states.append(env.current_state())
        pis.append(pi)
        movers.append(env.current_player())
        env.step(move)
        moves.append(int(move))
        ended, winner = env.game_end_winner()
        if ended:
            z = np.zeros(len(movers))
            if winner != -1:
                z[np.array(movers) == winner] = 1.0
                z[np.array(movers) != winner] = -1.0
            player.reset_player()
            return winner, list(zip(states, pis, z)), moves


def play_game(env, player1, player2):
    """game.py:61-94 (``start_player`` is ignored by the reference's reset, :73)."""
    env.reset()
    player1.set_player_id(0)
    player2.set_player_id(1)
    seats = {0: player1, 1: player2}
    moves = []
    while True:
        move = seats[env.current_player()].get_action(env)
        env.step(move)
        moves.append(int(move))
        ended, winner = env.game_end_winner()
        if ended:
            return winner, moves


def inverse_cdf_choice(uniforms):
    """numpy's legacy ``RandomState.choice(acts, p=probs)`` for one draw is
    ``acts[searchsorted(cumsum(p)/cumsum(p)[-1], u, side='right')]`` with ``u`` the next
    uniform of the stream; this returns a ``choice`` callable that consumes ``u`` from the
    given iterator instead, and records what it used in ``.used``."""
    it = iter(uniforms)

    def choice(acts, probs):
        cdf = np.cumsum(np.asarray(probs, dtype=np.float64))
        cdf /= cdf[-1]
        u = float(next(it))
        choice.used.append(u)
        return int(np.asarray(acts)[cdf.searchsorted(u, side='right')])

    choice.used = []
    return choice


def tree_dump(root, max_nodes=None):
    """Flatten a tree to {path tuple: (N, W)} over nodes with N > 0 (plus the root)."""
    out = {}
    stack = [((), root)]
    while stack:
        path, node = stack.pop()
        if node.n > 0 or not path:
            out[path] = (node.n, float(node.w))
        for a, kid in zip(node.acts, node.kids):
            if kid.n > 0:
                stack.append((path + (a, ), kid))
        if max_nodes is not None and len(out) >= max_nodes:
            break
    return out
