"""Batched MuZero self-play: many CartPole environments in lock-step on one GPU.

Per move (arXiv:1911.08265v2 appendix: ``play_game`` / ``run_mcts`` / ``select_action``):
  observation -> initial inference (representation + prediction) -> roots expanded with Dirichlet noise
  -> ``n_sims`` x [ rz_mz_select -> gather the parents' hidden states -> recurrent inference on the whole
     batch -> store the leaves' hidden states -> rz_mz_expand_backup ]
  -> action ~ visit counts ^ (1 / T) -> environment step -> the step goes to the episode's trajectory.
Everything between the observation and the chosen action stays on the device; the tree statistics live
in the HIP kernels, the small MLPs of the model run on PyTorch-ROCm (rocBLAS GEMMs)."""
import numpy as np


class Episode(object):
    """One finished (or running) episode: observations, actions, rewards, search policies, root values."""

    def __init__(self):
        self.obs, self.actions, self.rewards, self.policies, self.root_values = [], [], [], [], []

    @classmethod
    def from_arrays(cls, obs, actions, rewards, policies, root_values):
        """An episode cut out of the self-play history in one go: numpy arrays with the step as first axis."""
        ep = cls()
        ep.obs, ep.actions, ep.rewards, ep.policies, ep.root_values = obs, actions, rewards, policies, root_values
        return ep

    def __len__(self):
        return len(self.actions)


class ReplayBuffer(object):
    """Finished episodes + the target construction of the MuZero learner (``make_target``): K unrolled
    steps, n-step value targets bootstrapped from the stored root values."""

    def __init__(self, capacity=2000, unroll_steps=5, td_steps=10, discount=0.997, seed=0):
        self.capacity, self.K, self.td_steps, self.discount = capacity, unroll_steps, td_steps, discount
        self.episodes = []
        self.rng = np.random.RandomState(seed)

    def add(self, episode):
        if len(episode) == 0:
            return
        self.episodes.append(episode)
        if len(self.episodes) > self.capacity:
            self.episodes.pop(0)

    def __len__(self):
        return len(self.episodes)

    def _value_target(self, ep, t):
        boot = t + self.td_steps
        value = ep.root_values[boot] * self.discount ** self.td_steps if boot < len(ep.root_values) else 0.0
        for i, r in enumerate(ep.rewards[t:boot]):
            value += r * self.discount ** i
        return value

    def sample(self, batch_size, n_actions):
        """-> obs [b, obs_dim], actions [b, K], target value [b, K+1], target reward [b, K+1], target policy
        [b, K+1, A], mask [b, K+1] (0 past the end of the episode: no policy loss there)."""
        K = self.K
        eps = [self.episodes[i] for i in self.rng.randint(len(self.episodes), size=batch_size)]
        pos = [self.rng.randint(len(ep)) for ep in eps]
        obs = np.stack([ep.obs[t] for ep, t in zip(eps, pos)]).astype(np.float32)
        actions = np.zeros((batch_size, K), dtype=np.int64)
        tv = np.zeros((batch_size, K + 1), dtype=np.float32)
        tr = np.zeros((batch_size, K + 1), dtype=np.float32)
        tp = np.full((batch_size, K + 1, n_actions), 1.0 / n_actions, dtype=np.float32)
        mask = np.zeros((batch_size, K + 1), dtype=np.float32)
        for b, (ep, t) in enumerate(zip(eps, pos)):
            for k in range(K + 1):
                i = t + k
                if i < len(ep):
                    tv[b, k] = self._value_target(ep, i)
                    tp[b, k] = ep.policies[i]
                    mask[b, k] = 1.0
                if k > 0:
                    if i - 1 < len(ep):
                        actions[b, k - 1] = ep.actions[i - 1]
                        tr[b, k] = ep.rewards[i - 1]
                    else:
                        actions[b, k - 1] = self.rng.randint(n_actions)  # past the end: random action, zero targets
        return obs, actions, tv, tr, tp, mask


class MuZeroSelfPlay(object):

    def __init__(self, net, env, n_sims=50, discount=0.997, temperature=1.0, root_dirichlet_alpha=0.25,
                 root_exploration_fraction=0.25, seed=0, pb_c_base=19652.0, pb_c_init=1.25, use_graph=True, fused=None):
        """``fused``: run the whole search of a move in ONE kernel launch (csrc/rz_muzero.hip k_mz_search: the model is
        evaluated inside the kernel, weights and 64 games per workgroup resident in LDS); None = whenever the model fits
        it (hidden size 64, <= 8 actions).  Otherwise one hipGraph of ~15 launches per simulation (tree kernels +
        PyTorch-ROCm layers)."""
        import torch
        from .tree import MuZeroTree
        self.torch = torch
        self.net, self.env = net, env
        self.device = env.device
        self.n_envs, self.n_actions = env.n_envs, env.n_actions
        self.n_sims, self.discount, self.temperature = int(n_sims), float(discount), float(temperature)
        self.alpha, self.noise_frac = float(root_dirichlet_alpha), float(root_exploration_fraction)
        self.tree = MuZeroTree(self.n_envs, self.n_actions, self.n_sims, discount, pb_c_base, pb_c_init,
                               device=str(self.device))
        self.hidden = torch.zeros((self.n_envs, self.tree.slots_per_game, net.hidden), dtype=torch.float32,
                                  device=self.device)
        self.rows = torch.arange(self.n_envs, device=self.device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        # step history: rings [step % HIST][environment] on the host; an episode is cut out of them (one fancy index per
        # field) when it ends.  CartPole-v1 truncates at 500 steps, so 512 steps of history always cover an episode.
        self.HIST = max(512, int(getattr(env, 'max_episode_steps', 500)) + 12)
        G, A = self.n_envs, self.n_actions
        self._h_obs = np.zeros((self.HIST, G, net.obs_dim), dtype=np.float32)
        self._h_act = np.zeros((self.HIST, G), dtype=np.int64)
        self._h_rew = np.zeros((self.HIST, G), dtype=np.float64)
        self._h_pol = np.zeros((self.HIST, G, A), dtype=np.float32)
        self._h_val = np.zeros((self.HIST, G), dtype=np.float64)
        self._t = 0                                           # global step index of the next move
        self._ep_start = np.zeros(self.n_envs, dtype=np.int64)
        self.sims_done = 0
        self.moves_done = 0
        self.obs = env.observe()
        # One simulation (select -> gather -> recurrent inference -> scatter -> expand + backup) is a fixed
        # sequence of ~15 small launches with static shapes: captured once as a hipGraph and replayed n_sims
        # times per move.  Captured before any search (the capture runs the step on the still empty trees;
        # every search starts by re-initialising its roots).  Weight updates are in place: the graph stays valid.
        self.fused = (net.hidden == 64 and self.n_actions <= 8) if fused is None else bool(fused)
        self._model_seen = None
        if self.fused:
            use_graph = False
            self._refresh_model()
        self.search_events = None  # bench.py: HIP events around the fused search launches
        self._graph = None
        self.sim_events = None
        self.sim_step_label = ('k_mz_search (the whole %d-simulation search of a move in one launch: select, recurrent '
                               'inference on LDS-resident weights, expand + backup)' % self.n_sims) if self.fused else \
            'MuZero simulation step (k_mz_select + torch recurrent inference + k_mz_expand_backup, one hipGraph)'
        if use_graph:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(3):
                    self._sim_step()
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                self._sim_step()
            self._graph = graph

    def _refresh_model(self):
        """(Re-)upload the model's weights for the fused search if the torch module changed (a learner step)."""
        seen = tuple((p.data_ptr(), p._version) for p in self.net.parameters())
        if seen != self._model_seen:
            self.tree.load_model(self.net)
            self._model_seen = seen

    def _sim_step(self):
        t = self.torch
        parent, action, leaf = self.tree.select()
        state = self.hidden[self.rows, parent.long()]
        nxt, reward, logits, value = self.net.recurrent_inference(state, action.long())
        self.hidden[self.rows, leaf.long()] = nxt
        probs = t.softmax(logits, dim=1).contiguous()
        reward, value = reward.contiguous(), value.contiguous()
        self.tree.expand_backup(reward, probs, value)
        return parent, action, leaf, reward, probs, value

    def close(self):
        self.tree.close()

    # ------------------------------------------------------------------ search
    def search(self, obs, add_noise=True, record=None):
        """One MuZero search per environment from the observations ``obs`` [n, obs_dim].
        -> (visit counts int32 [n, A], root value float64 [n]).  ``record`` (list): receives per simulation
        (parent, action, leaf, reward, probs, value) tensors -- parity tests replay them through their CPython restatement."""
        t = self.torch
        tree, net = self.tree, self.net
        with t.no_grad():
            s0, logits, _ = net.initial_inference(obs)
            probs = t.softmax(logits, dim=1).contiguous()
            self.hidden[:, 0] = s0
            noise = None
            if add_noise and self.noise_frac > 0:
                conc = t.full((self.n_envs, self.n_actions), self.alpha, dtype=t.float64, device=self.device)
                g = t._standard_gamma(conc, generator=self.gen)
                noise = (g / g.sum(dim=1, keepdim=True)).contiguous()
            tree.init_roots(probs, noise, self.noise_frac)
            if record is not None:
                record.append(('root', probs.clone(), None if noise is None else noise.clone()))
            if self.fused:
                self._refresh_model()
                ev = None
                if self.search_events is not None:
                    ev = (t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True))
                    ev[0].record()
                trace = tree.search_fused(self.hidden, self.n_sims, trace=record is not None)
                if ev is not None:
                    ev[1].record()
                    self.search_events.append(ev)
                if record is not None:
                    for i_sim in range(self.n_sims):
                        record.append(tuple(trace[k][i_sim].clone() for k in ('parent', 'action', 'leaf', 'reward', 'probs', 'value')))
            for i_sim in range(0 if self.fused else self.n_sims):
                if self._graph is not None and record is None:
                    if self.sim_events is not None and i_sim % 8 == 3:  # bench.py: HIP events around a sample of the steps, on the launch stream
                        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                        a.record()
                        self._graph.replay()
                        b.record()
                        self.sim_events.append((a, b))
                        continue
                    self._graph.replay()
                    continue
                out = self._sim_step()
                if record is not None:
                    record.append(tuple(x.clone() for x in out))
            visits = tree.root_visits().clone()
            n, vsum, _, _ = tree.root_stats()
            root_value = vsum / n.clamp_min(1).to(t.float64)
        self.sims_done += self.n_sims * self.n_envs
        return visits, root_value

    def select_actions(self, visits):
        """pseudocode ``select_action``: sample from visit counts ^ (1 / temperature) (arg-max at T = 0)."""
        t = self.torch
        if self.temperature <= 0:
            return visits.argmax(dim=1)
        w = visits.to(t.float64) ** (1.0 / self.temperature)
        return t.multinomial(w / w.sum(dim=1, keepdim=True), 1, generator=self.gen).squeeze(1)

    # ------------------------------------------------------------------ game loop
    def play_move(self):
        """One move of every environment; returns the episodes that ended with it."""
        t = self.torch
        obs = self.obs
        visits, root_value = self.search(obs)
        actions = self.select_actions(visits)
        nxt_obs, reward, terminated, truncated = self.env.step(actions)
        # everything the host keeps of this move in ONE device-to-host copy: [obs | action | reward | visits | root value | done]
        D, A = obs.shape[1], self.n_actions
        packed = t.cat((obs.to(t.float64), actions.to(t.float64)[:, None], reward.to(t.float64)[:, None],
                        visits.to(t.float64), root_value.to(t.float64)[:, None],
                        (terminated | truncated).to(t.float64)[:, None]), dim=1).cpu().numpy()
        slot = self._t % self.HIST
        self._h_obs[slot] = packed[:, :D]
        self._h_act[slot] = packed[:, D].astype(np.int64)
        self._h_rew[slot] = packed[:, D + 1]
        vis = packed[:, D + 2:D + 2 + A]
        self._h_pol[slot] = (vis / vis.sum(axis=1, keepdims=True)).astype(np.float32)
        self._h_val[slot] = packed[:, D + 2 + A]
        done_h = packed[:, D + 3 + A] != 0.0
        self._t += 1
        finished = []
        ended = np.nonzero(done_h)[0]
        if len(ended):
            # all episodes that ended with this move in ONE gather per field, then split by length
            lengths = self._t - self._ep_start[ended]
            env_idx = np.repeat(ended, lengths)
            first = np.repeat(self._ep_start[ended], lengths)
            within = np.arange(int(lengths.sum())) - np.repeat(np.cumsum(lengths) - lengths, lengths)
            step_idx = (first + within) % self.HIST
            cuts = np.cumsum(lengths)[:-1]
            fields = [np.split(h[step_idx, env_idx], cuts) for h in (self._h_obs, self._h_act, self._h_rew, self._h_pol, self._h_val)]
            finished = [Episode.from_arrays(*parts) for parts in zip(*fields)]
            self._ep_start[ended] = self._t
        self.obs = nxt_obs
        self.moves_done += self.n_envs
        return finished

    def collect(self, n_moves):
        """``n_moves`` moves of every environment -> list of finished episodes."""
        out = []
        for _ in range(n_moves):
            out.extend(self.play_move())
        self.tree.check()
        return out
