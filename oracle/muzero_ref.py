"""Oracle: MuZero search and the CartPole-v1 dynamics (TEST INFRASTRUCTURE).

PARITY UNPINNED AGAINST THE REFERENCE: jianzhnie/RLZero only names MuZero (README.md:3,
rlzero/algorithms/rl_args.py:21-24) and ships no implementation, fixture or test of it
(SURVEY.md 8c, 8f rank 4).  What is restated here, line by line, is the PUBLISHED algorithm:

* search ........ Schrittwieser et al., "Mastering Atari, Go, Chess and Shogi by Planning with a
                  Learned Model", arXiv:1911.08265v2, appendix pseudocode: ``MinMaxStats``, ``Node``,
                  ``run_mcts``, ``select_child``, ``ucb_score``, ``expand_node``, ``backpropagate``,
                  ``add_exploration_noise``, ``select_action`` -- single-player form (``to_play`` is
                  constant, so ``backpropagate`` never flips the sign).  One stated difference: the
                  network hands over policy PROBABILITIES (its fp32 softmax) instead of logits, so
                  ``expand_node`` stores them instead of computing ``exp(logit) / sum``.
* environment ... CartPole-v1 as defined by Gymnasium (Farama-Foundation/Gymnasium,
                  ``gymnasium/envs/classic_control/cartpole.py``, v0.29: Barto, Sutton & Anderson's
                  equations, Euler integrator, tau = 0.02 s, |x| > 2.4 or |theta| > 12 degrees ends the
                  episode, reward 1 per step, 500-step time limit).  Gymnasium is not installed in
                  this image; the constants and the update order are restated from its published source.

Everything is CPython float64, one rounding per operation -- the arithmetic the HIP kernels
(rlzero_amd/csrc/rz_muzero.hip) reproduce bit for bit.
"""
import math

MAXIMUM_FLOAT_VALUE = float('inf')


class MinMaxStats(object):
    """A class that holds the min-max values of the tree (pseudocode ``MinMaxStats``, no known bounds)."""

    def __init__(self):
        self.maximum = -MAXIMUM_FLOAT_VALUE
        self.minimum = MAXIMUM_FLOAT_VALUE

    def update(self, value):
        self.maximum = max(self.maximum, value)
        self.minimum = min(self.minimum, value)

    def normalize(self, value):
        if self.maximum > self.minimum:
            return (value - self.minimum) / (self.maximum - self.minimum)
        return value


class Node(object):
    """pseudocode ``Node``; children are a list indexed by action."""
    __slots__ = ('visit_count', 'prior', 'value_sum', 'children', 'hidden_state', 'reward')

    def __init__(self, prior):
        self.visit_count = 0
        self.prior = prior
        self.value_sum = 0
        self.children = []
        self.hidden_state = None
        self.reward = 0

    def expanded(self):
        return len(self.children) > 0

    def value(self):
        if self.visit_count == 0:
            return 0
        return self.value_sum / self.visit_count


class MuZeroConfig(object):
    def __init__(self, num_simulations=50, discount=0.997, pb_c_base=19652, pb_c_init=1.25,
                 root_dirichlet_alpha=0.25, root_exploration_fraction=0.25):
        self.num_simulations = num_simulations
        self.discount = discount
        self.pb_c_base = pb_c_base
        self.pb_c_init = pb_c_init
        self.root_dirichlet_alpha = root_dirichlet_alpha
        self.root_exploration_fraction = root_exploration_fraction


def ucb_score(config, parent, child, min_max_stats):
    """pseudocode ``ucb_score`` (v2: the value score is reward + discount * value, normalised)."""
    pb_c = math.log((parent.visit_count + config.pb_c_base + 1) / config.pb_c_base) + config.pb_c_init
    pb_c *= math.sqrt(parent.visit_count) / (child.visit_count + 1)
    prior_score = pb_c * child.prior
    if child.visit_count > 0:
        value_score = min_max_stats.normalize(child.reward + config.discount * child.value())
    else:
        value_score = 0
    return prior_score + value_score


def select_child(config, node, min_max_stats):
    """pseudocode ``select_child``: ``max`` over (score, action, child) tuples -- on equal scores the
    LARGER action wins."""
    best = None
    for action, child in enumerate(node.children):
        key = (ucb_score(config, node, child, min_max_stats), action)
        if best is None or key > best[0]:
            best = (key, action, child)
    return best[1], best[2]


def expand_node(node, hidden_state, reward, policy_probs):
    """pseudocode ``expand_node`` with probabilities in place of exp(logits) / sum."""
    node.hidden_state = hidden_state
    node.reward = reward
    node.children = [Node(float(p)) for p in policy_probs]


def backpropagate(search_path, value, discount, min_max_stats):
    """pseudocode ``backpropagate``, single player."""
    for node in reversed(search_path):
        node.value_sum += value
        node.visit_count += 1
        min_max_stats.update(node.value())
        value = node.reward + discount * value


def add_exploration_noise(config, node, noise):
    """pseudocode ``add_exploration_noise`` with the Dirichlet sample supplied by the caller."""
    frac = config.root_exploration_fraction
    for child, n in zip(node.children, noise):
        child.prior = child.prior * (1 - frac) + n * frac


def run_mcts(config, root, recurrent_inference, log=None):
    """pseudocode ``run_mcts``.  ``recurrent_inference(hidden_state, action, path) ->
    (hidden_state', reward, policy_probs, value)``; ``path`` = the actions from the root (lets a test feed
    recorded network outputs back in).  Returns the MinMaxStats."""
    min_max_stats = MinMaxStats()
    for _ in range(config.num_simulations):
        node = root
        search_path = [node]
        actions = []
        while node.expanded():
            action, node = select_child(config, node, min_max_stats)
            actions.append(action)
            search_path.append(node)
        parent = search_path[-2]
        hidden, reward, probs, value = recurrent_inference(parent.hidden_state, actions[-1], tuple(actions))
        expand_node(node, hidden, float(reward), probs)
        if log is not None:
            log.append(tuple(actions))
        backpropagate(search_path, float(value), config.discount, min_max_stats)
    return min_max_stats


def tree_dump(root):
    """{path of actions: (N, value_sum, reward, prior)} over all nodes that exist."""
    out = {}
    stack = [((), root)]
    while stack:
        path, node = stack.pop()
        out[path] = (node.visit_count, float(node.value_sum), float(node.reward), float(node.prior))
        for a, child in enumerate(node.children):
            stack.append((path + (a, ), child))
    return out


# ------------------------------------------------------------------------- CartPole-v1
class RefCartPole(object):
    """Gymnasium ``CartPoleEnv`` (Euler) + the 500-step ``TimeLimit`` of CartPole-v1, scalar float64."""
    gravity = 9.8
    masscart = 1.0
    masspole = 0.1
    total_mass = masspole + masscart
    length = 0.5  # actually half the pole's length
    polemass_length = masspole * length
    force_mag = 10.0
    tau = 0.02  # seconds between state updates
    theta_threshold_radians = 12 * 2 * math.pi / 360
    x_threshold = 2.4
    max_episode_steps = 500

    def __init__(self):
        self.state = None
        self.steps = 0

    def reset(self, state):
        """Gymnasium draws the 4 components uniformly from [-0.05, 0.05]; the caller supplies them."""
        self.state = tuple(float(v) for v in state)
        self.steps = 0
        return self.state

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * (theta_dot * theta_dot) * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * (costheta * costheta) / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        self.steps += 1
        terminated = bool(x < -self.x_threshold or x > self.x_threshold or
                          theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians)
        truncated = self.steps >= self.max_episode_steps
        return self.state, 1.0, terminated, truncated
