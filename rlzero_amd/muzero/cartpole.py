"""CartPole-v1 for many environments at once, on the device (float64 state, float32 observations).

Gymnasium's ``CartPoleEnv`` (``gymnasium/envs/classic_control/cartpole.py``, v0.29: Euler integrator,
tau = 0.02 s, force 10 N, episode ends when |x| > 2.4 or |theta| > 12 degrees; reward 1 per step) with
the 500-step time limit of the ``CartPole-v1`` registration.  Gymnasium is not available offline; the
constants and the order of the updates are the published ones.  Finished environments are reset
in place (auto-reset), initial states are uniform in [-0.05, 0.05]^4 from a counter-based stream keyed
(seed, environment, episode), so a run does not depend on how environments are batched."""
import math

import numpy as np

GRAVITY, MASSCART, MASSPOLE, LENGTH, FORCE_MAG, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
TOTAL_MASS = MASSPOLE + MASSCART
POLEMASS_LENGTH = MASSPOLE * LENGTH
THETA_THRESHOLD = 12 * 2 * math.pi / 360
X_THRESHOLD = 2.4
MAX_EPISODE_STEPS = 500


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def initial_states(seed, env_ids, episodes):
    """float64 [n, 4] in [-0.05, 0.05): counter-based, reproducible on any host."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    episodes = np.asarray(episodes, dtype=np.uint64)
    with np.errstate(over='ignore'):
        key = _splitmix64(_splitmix64(np.uint64(seed) ^ (env_ids << np.uint64(24))) ^ episodes)
        out = np.empty((len(env_ids), 4), dtype=np.float64)
        for j in range(4):
            u = (_splitmix64(key ^ np.uint64(j + 1)) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
            out[:, j] = -0.05 + 0.1 * u
    return out


class CartPoleBatch(object):
    """The environments live on the device: ``state`` float64 [n, 4], ``steps`` / ``episode_dev`` int64 [n]; a step of
    all of them (physics, termination, auto-reset) is ONE kernel launch (csrc/rz_muzero.hip k_cartpole_step) and never
    synchronises with the host.  The fused MuZero moves (MuZeroTree.play_cartpole) step the same tensors in place."""
    n_actions = 2
    obs_dim = 4
    max_episode_steps = MAX_EPISODE_STEPS

    def __init__(self, n_envs, device='cuda:0', seed=0):
        import torch
        from .. import _hip
        self.torch = torch
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('CartPoleBatch runs on the GPU (device=%r)' % (device, ))
        self.lib = _hip.load()
        self.n_envs, self.seed = int(n_envs), int(seed)
        kw = dict(device=self.device)
        self.state = torch.zeros((n_envs, 4), dtype=torch.float64, **kw)
        self.steps = torch.zeros(n_envs, dtype=torch.int64, **kw)
        self.episode_dev = torch.zeros(n_envs, dtype=torch.int64, **kw)
        self._obs = torch.zeros((n_envs, 4), dtype=torch.float32, **kw)
        self._reward = torch.zeros(n_envs, dtype=torch.float32, **kw)
        self._terminated = torch.zeros(n_envs, dtype=torch.uint8, **kw)
        self._truncated = torch.zeros(n_envs, dtype=torch.uint8, **kw)
        self.reset_all()

    @property
    def episode(self):
        """Episode index of every environment (host copy)."""
        return self.episode_dev.cpu().numpy()

    def reset_all(self):
        self.set_states(initial_states(self.seed, np.arange(self.n_envs), self.episode))

    def set_states(self, states, mask=None):
        t = self.torch
        new = t.from_numpy(np.ascontiguousarray(states, dtype=np.float64)).to(self.device)
        if mask is None:
            self.state.copy_(new)
            self.steps.zero_()
        else:
            m = t.from_numpy(np.ascontiguousarray(mask, dtype=np.bool_)).to(self.device)
            self.state[m] = new[m]
            self.steps[m] = 0

    def observe(self):
        return self.state.to(self.torch.float32)

    def step(self, actions):
        """actions int64 [n] on the device -> (obs float32 [n,4] AFTER auto-reset, reward float32 [n],
        terminated bool [n], truncated bool [n]) ; the observation of a finished environment is the first of
        its next episode (the terminal state itself is never evaluated by MuZero)."""
        import ctypes
        from ..engine import _ptr, check
        t = self.torch
        actions = actions.to(t.int64).contiguous()
        stream = ctypes.c_void_p(t.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.rz_cartpole_step(_ptr(self.state), _ptr(self.steps), _ptr(self.episode_dev), _ptr(actions), self.n_envs,
                                        self.seed & (2 ** 64 - 1), _ptr(self._obs), _ptr(self._reward), _ptr(self._terminated),
                                        _ptr(self._truncated), stream), 'rz_cartpole_step')
        return self._obs.clone(), self._reward.clone(), self._terminated.bool(), self._truncated.bool()
