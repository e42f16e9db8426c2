"""Oracle: leaf evaluators (TEST INFRASTRUCTURE, see oracle/__init__.py).

* ``v0`` / ``vlin``: synthetic, exactly representable evaluators defined in SURVEY.md
  Appendix B; they exist so tree parity does not depend on last-bit differences between
  CPU and GPU convolutions.  Both take any env with the reference's interface
  (``leagel_actions()``, ``states``, ``current_player()``).
* ``net_forward`` / ``NetEvaluator``: functional restatement of
  ``rlzero/games/gomoku/policy_value_net.py:34-52`` and of the batch-1 evaluator
  ``rlzero/games/gomoku/alphazero_agent.py:31-46`` on torch CPU tensors.
* ``numpy_weights``: deterministic weights from ``numpy.random.RandomState`` (MT19937 is
  stable across numpy versions) so net fixtures need not store parameters.
"""
import numpy as np


def v0(env):
    legal = env.leagel_actions()
    k = len(legal)
    priors = np.ones(k) / k if k else np.ones(0)
    return list(zip(legal, priors)), 0.0


def vlin_value(states, current_player):
    s = 0
    for a, p in states.items():
        s += (a + 1) * (1 if p == 0 else 3)
    s += 5 * current_player
    return ((s % 17) - 8) / 8.0


def vlin(env):
    legal = env.leagel_actions()
    k = len(legal)
    priors = np.ones(k) / k if k else np.ones(0)
    return list(zip(legal, priors)), vlin_value(env.states, env.current_player())


# --------------------------------------------------------------------------------------
PARAM_SHAPES = (
    # name, shape builder(board S=B*B)  -- creation order of policy_value_net.py:12-25
    ('conv1.weight', lambda S: (32, 4, 3, 3)),
    ('conv1.bias', lambda S: (32, )),
    ('conv2.weight', lambda S: (64, 32, 3, 3)),
    ('conv2.bias', lambda S: (64, )),
    ('conv3.weight', lambda S: (128, 64, 3, 3)),
    ('conv3.bias', lambda S: (128, )),
    ('act_conv1.weight', lambda S: (4, 128, 1, 1)),
    ('act_conv1.bias', lambda S: (4, )),
    ('act_fc1.weight', lambda S: (S, 4 * S)),
    ('act_fc1.bias', lambda S: (S, )),
    ('val_conv1.weight', lambda S: (2, 128, 1, 1)),
    ('val_conv1.bias', lambda S: (2, )),
    ('val_fc1.weight', lambda S: (64, 2 * S)),
    ('val_fc1.bias', lambda S: (64, )),
    ('val_fc2.weight', lambda S: (1, 64)),
    ('val_fc2.bias', lambda S: (1, )),
)


def numpy_weights(board_size, seed):
    """fp32 parameters drawn as U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from RandomState(seed),
    in PARAM_SHAPES order.  Returns {name: np.float32 array}."""
    rs = np.random.RandomState(seed)
    S = board_size * board_size
    out = {}
    fan_in = 1
    for name, shape_of in PARAM_SHAPES:
        shape = shape_of(S)
        if name.endswith('.weight'):
            fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / np.sqrt(fan_in)
        out[name] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def net_forward(weights, obs, dtype=None):
    """policy_value_net.py:34-52 on torch CPU.  ``weights``: {name: array/tensor};
    ``obs``: [b,4,B,B].  Returns (log_probs [b,S], value [b,1]) as torch tensors."""
    import torch
    import torch.nn.functional as F
    dtype = dtype or torch.float32
    w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}
    x = torch.as_tensor(np.asarray(obs)).to(dtype)
    b = x.shape[0]
    x = F.relu(F.conv2d(x, w['conv1.weight'], w['conv1.bias'], padding=1))
    x = F.relu(F.conv2d(x, w['conv2.weight'], w['conv2.bias'], padding=1))
    x = F.relu(F.conv2d(x, w['conv3.weight'], w['conv3.bias'], padding=1))
    a = F.relu(F.conv2d(x, w['act_conv1.weight'], w['act_conv1.bias']))
    a = F.linear(a.reshape(b, -1), w['act_fc1.weight'], w['act_fc1.bias'])
    logp = F.log_softmax(a, dim=1)
    v = F.relu(F.conv2d(x, w['val_conv1.weight'], w['val_conv1.bias']))
    v = F.relu(F.linear(v.reshape(b, -1), w['val_fc1.weight'], w['val_fc1.bias']))
    v = torch.tanh(F.linear(v, w['val_fc2.weight'], w['val_fc2.bias']))
    return logp, v


class NetEvaluator(object):
    """alphazero_agent.py:31-46: batch-1 forward, p = exp(log_softmax)[legal] (not
    renormalised), v = python float of the fp32 value."""

    def __init__(self, weights, board_size):
        self.weights = weights
        self.board_size = board_size
        self.n_calls = 0

    def __call__(self, env):
        import torch
        legal = env.leagel_actions()
        rows, cols = self.board_size if isinstance(self.board_size, (tuple, list)) else (self.board_size, self.board_size)
        obs = env.current_state().reshape(-1, 4, rows, cols)
        with torch.no_grad():
            logp, v = net_forward(self.weights, np.ascontiguousarray(obs))
        probs = np.exp(logp.numpy().flatten())
        self.n_calls += 1
        return list(zip(legal, probs[legal])), v.item()
