"""ctypes binding of the MuZero search tree kernels (include/rlzero_hip.h, rz_mz_*)."""
import ctypes

import numpy as np

from .. import _hip
from ..engine import _ptr, check


class MuZeroTree(object):
    """Search trees of ``n_games`` environments on one GPU.  Step by step (``select`` / ``expand_backup``) the learned
    model is the caller's: see ``MuZeroSelfPlay.search`` for the simulation loop; ``search_fused`` and ``play_cartpole``
    evaluate it inside the kernel (``load_model`` / ``load_representation`` first)."""

    def __init__(self, n_games, n_actions, n_sims, discount=0.997, pb_c_base=19652.0, pb_c_init=1.25,
                 device='cuda:0'):
        import torch
        self.torch = torch
        self.lib = _hip.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda' or not torch.cuda.is_available():
            raise _hip.HipError('MuZeroTree needs an MI355X (device=%r): there is no CPU path' % (device, ))
        self.n_games, self.n_actions, self.n_sims = int(n_games), int(n_actions), int(n_sims)
        self.discount = float(discount)
        cfg = _hip.RzMzConfig(_hip.ABI_VERSION, self.n_games, self.n_actions, self.n_sims, self.discount,
                              float(pb_c_base), float(pb_c_init), self.device.index or 0, 0)
        self.handle = ctypes.c_void_p()
        check(self.lib.rz_mz_create(ctypes.byref(cfg), ctypes.byref(self.handle)), 'rz_mz_create')
        slots, nbytes = ctypes.c_int32(0), ctypes.c_int64(0)
        check(self.lib.rz_mz_geometry(self.handle, ctypes.byref(slots), ctypes.byref(nbytes)), 'rz_mz_geometry')
        self.slots_per_game, self.device_bytes = slots.value, nbytes.value
        kw = dict(device=self.device)
        G, A = self.n_games, self.n_actions
        self.parent = torch.zeros(G, dtype=torch.int32, **kw)
        self.action = torch.zeros(G, dtype=torch.int32, **kw)
        self.leaf = torch.zeros(G, dtype=torch.int32, **kw)
        self.visits = torch.zeros((G, A), dtype=torch.int32, **kw)
        self.child_f64 = torch.zeros((G, A), dtype=torch.float64, **kw)
        self.root_n = torch.zeros(G, dtype=torch.int32, **kw)
        self.root_sum = torch.zeros(G, dtype=torch.float64, **kw)
        self.vmin = torch.zeros(G, dtype=torch.float64, **kw)
        self.vmax = torch.zeros(G, dtype=torch.float64, **kw)
        # the very table CPython's math.log gives on this host
        import math
        tab = np.array([math.log((n + pb_c_base + 1) / pb_c_base) for n in range(self.n_sims + 2)], dtype=np.float64)
        check(self.lib.rz_mz_upload_log_table(self.handle, ctypes.c_void_p(tab.ctypes.data), tab.size),
              'rz_mz_upload_log_table')

    MODEL_PARAMS = ('dyn1.weight', 'dyn1.bias', 'dyn2.weight', 'dyn2.bias', 'rew1.weight', 'rew1.bias', 'rew2.weight',
                    'rew2.bias', 'pre1.weight', 'pre1.bias', 'pol.weight', 'pol.bias', 'val.weight', 'val.bias')

    def load_model(self, net):
        """Upload the dynamics / reward / prediction layers of a MuZeroNet (hidden size 64) for ``search_fused``; call
        again after every optimiser step."""
        sd = net.state_dict()
        arrays = [sd[name].detach().to('cpu', self.torch.float32).contiguous().numpy() for name in self.MODEL_PARAMS]
        ptrs = (ctypes.c_void_p * 14)(*[a.ctypes.data for a in arrays])
        check(self.lib.rz_mz_load_model(self.handle, ptrs, 14, int(net.hidden)), 'rz_mz_load_model')

    REPRESENTATION_PARAMS = ('rep1.weight', 'rep1.bias', 'rep2.weight', 'rep2.bias')

    def load_representation(self, net):
        """Upload the representation layers h(o) of a MuZeroNet (for ``play_cartpole``); call again after every
        optimiser step, like ``load_model``."""
        sd = net.state_dict()
        arrays = [sd[name].detach().to('cpu', self.torch.float32).contiguous().numpy() for name in self.REPRESENTATION_PARAMS]
        ptrs = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in arrays])
        check(self.lib.rz_mz_load_representation(self.handle, ptrs, 4, int(net.obs_dim), int(net.hidden)),
              'rz_mz_load_representation')

    def play_cartpole(self, hidden, n_sims, n_moves, env, noise_seed, noise_frac, alpha, temperature, history, first_step,
                      arena, counters, entries):
        """``n_moves`` whole moves of the CartPoleBatch ``env`` in ONE launch (k_mz_search, MOVES stages): initial
        inference, root noise, ``n_sims`` simulations, action from the visit counts, environment step.  ``history`` =
        (ring float64 [G, ring_steps, 8 + A], episode_start int64 [G]): every move's record -- observation (4) | action |
        reward | visits (A) | root value | done -- lands in the ring at step % ring_steps (step = first_step + move); the
        records of every episode that ENDS are copied as one run into ``arena`` float64 [rows, 8 + A] and described in
        ``entries`` int64 [n, 4] (environment, end step, length, first arena row or -1); ``counters`` int64 [4] (zeroed
        here): arena rows claimed, entries, episodes that did not fit."""
        ring, episode_start = history
        counters.zero_()
        play = _hip.RzMzCartPolePlay(
            env.state.data_ptr(), env.steps.data_ptr(), env.episode_dev.data_ptr(), episode_start.data_ptr(),
            int(env.seed) & (2 ** 64 - 1), int(noise_seed) & (2 ** 64 - 1), float(noise_frac), float(alpha), float(temperature),
            ring.data_ptr(), int(ring.shape[1]), 0, int(first_step), arena.data_ptr(), int(arena.shape[0]),
            counters.data_ptr(), entries.data_ptr(), int(entries.shape[0]))
        check(self.lib.rz_mz_play_cartpole(self.handle, _ptr(hidden), int(n_sims), int(n_moves), ctypes.byref(play), self.stream()),
              'rz_mz_play_cartpole')

    def set_search_shape(self, games_per_workgroup=0):
        """Games per workgroup of ``search_fused`` (<= 16; 0 = chosen from the number of games and CUs)."""
        check(self.lib.rz_mz_set_search_shape(self.handle, int(games_per_workgroup)), 'rz_mz_set_search_shape')
        return self

    def search_fused(self, hidden, n_sims, trace=False):
        """All ``n_sims`` simulations of every game in ONE launch (k_mz_search; init_roots and the root's hidden state
        in ``hidden[:, 0]`` first).  ``trace``: returns per simulation what the kernel selected and its network outputs:
        dict of tensors parent / action / leaf int32 [n_sims, G], reward / value float32 [n_sims, G], probs [n_sims, G, A]."""
        t = self.torch
        out = None
        args = [None] * 6
        if trace:
            kw = dict(device=self.device)
            G, A = self.n_games, self.n_actions
            out = {'parent': t.zeros((n_sims, G), dtype=t.int32, **kw), 'action': t.zeros((n_sims, G), dtype=t.int32, **kw),
                   'leaf': t.zeros((n_sims, G), dtype=t.int32, **kw), 'reward': t.zeros((n_sims, G), dtype=t.float32, **kw),
                   'probs': t.zeros((n_sims, G, A), dtype=t.float32, **kw), 'value': t.zeros((n_sims, G), dtype=t.float32, **kw)}
            args = [_ptr(out[k]) for k in ('parent', 'action', 'leaf', 'reward', 'probs', 'value')]
        check(self.lib.rz_mz_search(self.handle, _ptr(hidden), int(n_sims), *args, self.stream()), 'rz_mz_search')
        return out

    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if self.handle:
            self.lib.rz_mz_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mask(self, mask):
        return _ptr(mask) if mask is not None else None

    def init_roots(self, probs, noise=None, noise_frac=0.25, mask=None):
        """probs float32 [G, A] (device); noise float64 [G, A] Dirichlet samples or None; mask uint8 [G]."""
        check(self.lib.rz_mz_init_roots(self.handle, _ptr(probs), _ptr(noise) if noise is not None else None,
                                        float(noise_frac), self._mask(mask), self.stream()), 'rz_mz_init_roots')

    def select(self, mask=None):
        check(self.lib.rz_mz_select(self.handle, _ptr(self.parent), _ptr(self.action), _ptr(self.leaf),
                                    self._mask(mask), self.stream()), 'rz_mz_select')
        return self.parent, self.action, self.leaf

    def expand_backup(self, reward, probs, value, mask=None):
        check(self.lib.rz_mz_expand_backup(self.handle, _ptr(reward), _ptr(probs), _ptr(value), self._mask(mask),
                                           self.stream()), 'rz_mz_expand_backup')

    def root_visits(self):
        check(self.lib.rz_mz_root_children(self.handle, 0, _ptr(self.visits), self.stream()), 'rz_mz_root_children')
        return self.visits

    def root_children(self, what):
        """'value_sum' | 'reward' | 'prior' of the root's children, float64 [G, A]."""
        code = {'value_sum': 1, 'reward': 2, 'prior': 3}[what]
        check(self.lib.rz_mz_root_children(self.handle, code, _ptr(self.child_f64), self.stream()),
              'rz_mz_root_children')
        return self.child_f64.clone()

    def root_stats(self):
        """-> (N, value_sum, min, max) of the roots / their MinMaxStats."""
        check(self.lib.rz_mz_root_stats(self.handle, _ptr(self.root_n), _ptr(self.root_sum), _ptr(self.vmin),
                                        _ptr(self.vmax), self.stream()), 'rz_mz_root_stats')
        return self.root_n, self.root_sum, self.vmin, self.vmax

    def check(self):
        flags = ctypes.c_int32(0)
        check(self.lib.rz_mz_error_flags(self.handle, ctypes.byref(flags)), 'rz_mz_error_flags')
        if flags.value:
            raise _hip.HipError('MuZero tree error flags 0x%x (tree arena full)' % flags.value)
