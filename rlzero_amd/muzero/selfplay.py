"""Batched MuZero self-play: many CartPole environments in lock-step on one GPU.

Per move (arXiv:1911.08265v2 appendix: ``play_game`` / ``run_mcts`` / ``select_action``):
  observation -> initial inference (representation + prediction) -> roots expanded with Dirichlet noise
  -> ``n_sims`` x [ rz_mz_select -> gather the parents' hidden states -> recurrent inference on the whole
     batch -> store the leaves' hidden states -> rz_mz_expand_backup ]
  -> action ~ visit counts ^ (1 / T) -> environment step -> the step goes to the episode's trajectory.
Three routes (``MuZeroSelfPlay(fused=..., fused_moves=...)``): the step-by-step one above (tree kernels + the model's
small MLPs on PyTorch-ROCm, one hipGraph per simulation); the search of a move in ONE launch with the model evaluated
inside the kernel (``rz_mz_search``); and whole MOVES in one launch -- initial inference, noise, search, action draw,
CartPole step, episode history on the device (``rz_mz_play_cartpole``), the host reading finished episodes a launch
behind the GPU.  The last is the default for a ``CartPoleBatch`` environment and a hidden size of 64."""
import numpy as np


class Episode(object):
    """One finished (or running) episode: observations, actions, rewards, search policies, root values."""
    FIELDS = ('obs', 'actions', 'rewards', 'policies', 'root_values')

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


class EpisodeSeq(object):
    """The episodes that ended during a stretch of self-play, as a sequence: the steps of all of them lie in a few flat
    arrays (one block per stretch) and an ``Episode`` -- views into those arrays -- is made when it is asked for.
    With thousands of environments hundreds of episodes end per move: building them eagerly was the host's largest cost."""

    def __init__(self):
        self._chunks = []   # (fields tuple of flat arrays, first row int64 [n], lengths int64 [n])
        self._starts = [0]  # index of the first episode of every chunk (+ the total)

    def _add(self, fields, lengths, first_rows=None):
        lengths = np.asarray(lengths, dtype=np.int64)
        if first_rows is None:  # the episodes lie one behind the other
            first_rows = np.cumsum(lengths) - lengths
        self._chunks.append((fields, np.asarray(first_rows, dtype=np.int64), lengths))
        self._starts.append(self._starts[-1] + len(lengths))

    def extend(self, other):
        for chunk in other._chunks:
            self._chunks.append(chunk)
            self._starts.append(self._starts[-1] + len(chunk[2]))

    def __len__(self):
        return self._starts[-1]

    def _add_packed(self, block, obs_dim, n_actions, lengths, first_rows=None):
        """Episodes as rows of the device's packed records -- float64 [rows, obs | action | reward | visit counts | root value]
        (rz_mz_play_cartpole) -- kept as they came: the fields of an episode are formed when it is asked for.  (Forming them
        for every launch's 130 k rows cost the host about as long as the launch takes the GPU.)"""
        lengths = np.asarray(lengths, dtype=np.int64)
        if first_rows is None:
            first_rows = np.cumsum(lengths) - lengths
        self._chunks.append(((block, int(obs_dim), int(n_actions)), np.asarray(first_rows, dtype=np.int64), lengths))
        self._starts.append(self._starts[-1] + len(lengths))

    @staticmethod
    def fields_of_packed(block, D, A):
        vis = block[:, D + 2:D + 2 + A]
        pol = (vis / vis.sum(axis=1, keepdims=True)).astype(np.float32)
        return (block[:, :D].astype(np.float32), block[:, D].astype(np.int64), block[:, D + 1].copy(), pol, block[:, D + 2 + A].copy())

    def _make(self, chunk, j):
        fields, first_rows, lengths = self._chunks[chunk]
        a = int(first_rows[j])
        b = a + int(lengths[j])
        if len(fields) == 3 and isinstance(fields[1], int):   # packed records: cut the episode's rows out, then form its fields
            return Episode.from_arrays(*self.fields_of_packed(fields[0][a:b], fields[1], fields[2]))
        return Episode.from_arrays(*(f[a:b] for f in fields))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        chunk = int(np.searchsorted(self._starts, i, side='right')) - 1
        return self._make(chunk, i - self._starts[chunk])

    def __iter__(self):
        for chunk, (_, _, lengths) in enumerate(self._chunks):
            for j in range(len(lengths)):
                yield self._make(chunk, j)

    def lengths(self):
        """Number of steps of every episode, int64 [len(self)]."""
        return np.concatenate([c[2] for c in self._chunks]) if self._chunks else np.zeros(0, dtype=np.int64)


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
        # an Episode cut out of an EpisodeSeq is views into that stretch's block: kept as they are, one stored episode would
        # keep the whole block (thousands of episodes) alive -- the buffer owns its copies
        episode = Episode.from_arrays(*(np.array(getattr(episode, name)) for name in Episode.FIELDS))
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
                 root_exploration_fraction=0.25, seed=0, pb_c_base=19652.0, pb_c_init=1.25, use_graph=True, fused=None,
                 fused_moves=None, moves_per_launch=16, arena_rows=None):
        """``fused``: run the whole search of a move in ONE kernel launch (csrc/rz_muzero.hip k_mz_search: the model is
        evaluated inside the kernel on the matrix pipe, 16 games per workgroup, trees in LDS); None = whenever the model
        fits it (hidden size 64, <= 8 actions).  Otherwise one hipGraph of ~15 launches per simulation (tree kernels +
        PyTorch-ROCm layers).
        ``fused_moves``: whole MOVES in one launch (the MOVES stages of k_mz_search: initial inference, root noise, search,
        action draw, environment step; ``moves_per_launch`` of them per launch) -- the host only reads one packed record
        per environment and move, a chunk behind the GPU.  None = whenever ``fused`` holds and the environment is a
        CartPoleBatch (the environment step is device code).  Its random draws (root noise, actions) come from the
        kernel's counter-based stream keyed (seed, environment, episode, step), not from the torch generator.
        ``arena_rows``: records the per-launch arena of finished episodes holds (None = twice what a launch plays + a few
        long episodes; episodes that do not fit are read back from the device ring instead)."""
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
        # step history of the host-driven loop: rings [step % HIST][environment] on the host (allocated at the first
        # move: the fused moves keep their history on the device); an episode is cut out of them (one fancy index per
        # field) when it ends.  CartPole-v1 truncates at 500 steps, so 512 steps of history always cover an episode.
        if not hasattr(env, 'max_episode_steps'):
            raise ValueError('the environment must state max_episode_steps: the step history is a ring of that many steps')
        self.HIST = max(512, int(env.max_episode_steps) + 12)
        self._h_obs = None
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
        from .cartpole import CartPoleBatch
        can_fuse_moves = self.fused and isinstance(env, CartPoleBatch) and net.obs_dim == 4 and self.n_actions == 2
        self.fused_moves = can_fuse_moves if fused_moves is None else bool(fused_moves)
        if self.fused_moves and not can_fuse_moves:
            raise ValueError('fused_moves needs the fused search and a CartPoleBatch environment')
        self.moves_per_launch = max(1, int(moves_per_launch))
        self._arena_rows = arena_rows
        self.noise_seed = int(seed)
        self._records = None  # fused moves: [device records, pinned host copy, event] x 3 (two launches ahead of the host)
        if self.fused:
            use_graph = False
            self._refresh_model()
        self.search_events = None  # bench.py: HIP events around the fused search launches
        self._graph = None
        self.sim_events = None
        if self.fused_moves:
            self.sim_step_label = ('k_mz_search with its MOVES stages (whole moves in one launch: initial inference, root noise, '
                                   '%d simulations -- select, recurrent inference on the matrix pipe, expand + backup -- action draw, '
                                   'CartPole step)' % self.n_sims)
        elif self.fused:
            self.sim_step_label = ('k_mz_search (the whole %d-simulation search of a move in one launch: select, recurrent '
                                   'inference on the matrix pipe with register-resident weights, expand + backup)' % self.n_sims)
        else:
            self.sim_step_label = 'MuZero simulation step (k_mz_select + torch recurrent inference + k_mz_expand_backup, one hipGraph)'
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
            if self.fused_moves:
                self.tree.load_representation(self.net)
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
        if self.fused_moves:
            finished = self._collect_fused(1)
            self.obs = self.env.observe()
            return finished
        t = self.torch
        obs = self.obs
        visits, root_value = self.search(obs)
        actions = self.select_actions(visits)
        nxt_obs, reward, terminated, truncated = self.env.step(actions)
        # everything the host keeps of this move in ONE device-to-host copy: [obs | action | reward | visits | root value | done]
        packed = t.cat((obs.to(t.float64), actions.to(t.float64)[:, None], reward.to(t.float64)[:, None],
                        visits.to(t.float64), root_value.to(t.float64)[:, None],
                        (terminated | truncated).to(t.float64)[:, None]), dim=1).cpu().numpy()
        finished = self._ingest(packed)
        self.obs = nxt_obs
        return finished

    def _ingest(self, rec):
        """The host's part of self-play: the packed records of K consecutive moves (float64 [K, n, obs | action | reward |
        visits | root value | done]; [n, row] = one move) into the history rings; -> the episodes that ended during them,
        in the order of their last move (then environment).  One pass of array operations for the whole stretch: with
        thousands of environments the per-call overhead of a move-by-move loop was the bottleneck of self-play."""
        rec = rec[None] if rec.ndim == 2 else rec
        K = rec.shape[0]
        D, A = self.net.obs_dim, self.n_actions
        if self._h_obs is None:
            G = self.n_envs
            self.HIST = max(self.HIST, K + 512)
            self._h_obs = np.zeros((self.HIST, G, D), dtype=np.float32)
            self._h_act = np.zeros((self.HIST, G), dtype=np.int64)
            self._h_rew = np.zeros((self.HIST, G), dtype=np.float64)
            self._h_pol = np.zeros((self.HIST, G, A), dtype=np.float32)
            self._h_val = np.zeros((self.HIST, G), dtype=np.float64)
        t0 = self._t
        slots = (t0 + np.arange(K)) % self.HIST
        self._h_obs[slots] = rec[:, :, :D]
        self._h_act[slots] = rec[:, :, D]
        self._h_rew[slots] = rec[:, :, D + 1]
        vis = rec[:, :, D + 2:D + 2 + A]
        self._h_pol[slots] = vis / vis.sum(axis=2, keepdims=True)
        self._h_val[slots] = rec[:, :, D + 2 + A]
        self._t += K
        self.moves_done += self.n_envs * K
        finished = EpisodeSeq()
        kk, ee = np.nonzero(rec[:, :, D + 3 + A] != 0.0)   # (move, environment) of every episode end, by move then environment
        if len(kk):
            end = t0 + kk + 1                                # global step index behind the episode's last move
            # an environment may finish more than once in the stretch: its later episodes start where the previous ended
            order = np.lexsort((kk, ee))
            ee_s, end_s = ee[order], end[order]
            first = np.ones(len(order), dtype=bool)
            first[1:] = ee_s[1:] != ee_s[:-1]
            start_s = np.empty_like(end_s)
            start_s[first] = self._ep_start[ee_s[first]]
            start_s[~first] = end_s[:-1][~first[1:]]
            last = np.ones(len(order), dtype=bool)
            last[:-1] = first[1:]
            self._ep_start[ee_s[last]] = end_s[last]
            start = np.empty_like(start_s)
            start[order] = start_s
            # all episodes in ONE gather per field; an Episode is cut out when it is read
            lengths = end - start
            if int(lengths.max()) > self.HIST:   # (cannot happen while the environment keeps its max_episode_steps)
                raise RuntimeError('an episode of %d steps does not fit the history ring of %d' % (int(lengths.max()), self.HIST))
            env_idx = np.repeat(ee, lengths)
            within = np.arange(int(lengths.sum())) - np.repeat(np.cumsum(lengths) - lengths, lengths)
            step_idx = (np.repeat(start, lengths) + within) % self.HIST
            finished._add(tuple(h[step_idx, env_idx] for h in (self._h_obs, self._h_act, self._h_rew, self._h_pol, self._h_val)),
                          lengths)
        return finished

    # ------------------------------------------------------------------ whole moves on the device
    def _fused_state(self):
        """Device-side history of the fused moves (MuZeroTree.play_cartpole) and the two sets of per-launch buffers."""
        if self._records is None:
            t = self.torch
            G, K = self.n_envs, self.moves_per_launch
            row = self.net.obs_dim + 4 + self.n_actions
            kw = dict(device=self.device)
            steps = int(self.env.max_episode_steps) + 2 * K + 12
            self._ring = t.zeros((G, steps, row), dtype=t.float64, **kw)
            self._ep_start_dev = t.full((G, ), self._t, dtype=t.int64, **kw)
            # in the steady state a launch ends about as many steps of episodes as it plays (G x K); but environments that start
            # together also END together for a while (8192 random-policy episodes of ~22 steps: whole launches in which 1.5 x G x K
            # rows end), so the arena holds twice that, plus a few long episodes -- what does not fit is read back from the ring
            # (entry with row -1), a slow path: a gather and a blocking copy per launch (an arena of 1.25 x G x K, tried in round 4,
            # took it in every launch of the bench: 5 ms each)
            rows, n_entries = 2 * G * K + 4 * steps, G * K
            if self._arena_rows is not None:
                rows = max(1, int(self._arena_rows))
            self._copy_stream = t.cuda.Stream(device=self.device)
            self._records = []
            for _ in range(3):   # two launches in flight + the one the host reads
                dev = (t.zeros(4, dtype=t.int64, **kw), t.zeros((n_entries, 4), dtype=t.int64, **kw),
                       t.zeros((rows, row), dtype=t.float64, **kw))
                host = tuple(t.zeros(x.shape, dtype=x.dtype).pin_memory() for x in dev)
                self._records.append((dev, host, t.cuda.Event()))
        return self._records

    def device_history(self):
        """(ring float64 [n_envs, ring_steps, obs | action | reward | visits | root value | done], episode_start int64
        [n_envs]) of the fused moves as numpy arrays: every move still in the ring, finished or not (tests, debugging)."""
        self._fused_state()
        return self._ring.cpu().numpy(), self._ep_start_dev.cpu().numpy()

    def _launch_moves(self, n_moves, buf):
        """Enqueue ``n_moves`` moves of every environment (one launch) and the copy of what ended during them."""
        t = self.torch
        (counters, entries, arena), host, event = buf
        with t.no_grad():
            self._refresh_model()
            ev = None
            if self.search_events is not None:
                ev = (t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True), n_moves)
                ev[0].record()
            self.tree.play_cartpole(self.hidden, self.n_sims, n_moves, self.env, self.noise_seed, self.noise_frac, self.alpha,
                                    self.temperature, (self._ring, self._ep_start_dev), self._t, arena, counters, entries)
            if ev is not None:
                ev[1].record()
                self.search_events.append(ev)
            # the copies run on their own stream: the next launch (other set of buffers) does not wait behind them
            done = t.cuda.Event()
            done.record()
            with t.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(done)
                for h, d in zip(host, (counters, entries, arena)):
                    h.copy_(d, non_blocking=True)
                event.record()
        self._t += n_moves
        self.sims_done += self.n_sims * self.n_envs * n_moves
        self.moves_done += self.n_envs * n_moves

    def _episodes_of_launch(self, buf):
        """The finished episodes of a launch from its arena (host copy), ordered by (last move, environment)."""
        _, (counters, entries, arena), event = buf
        event.synchronize()
        D, A = self.net.obs_dim, self.n_actions
        n_rows, n_entries, missed = (int(v) for v in counters.numpy()[:3])
        finished = EpisodeSeq()
        if n_entries == 0:
            return finished
        ent = entries.numpy()[:n_entries]
        ent = ent[np.lexsort((ent[:, 0], ent[:, 1]))]
        fits = ent[:, 3] >= 0

        if fits.any():
            used = min(n_rows, arena.shape[0])
            # (a copy: the pinned buffer is the next-but-one launch's; the episodes' fields are formed when they are asked for)
            finished._add_packed(arena.numpy()[:used].copy(), D, A, ent[fits, 2], ent[fits, 3])
        if missed:  # the arena was too small for these: their records are still in the ring
            t = self.torch
            env, end, length = (ent[~fits, c] for c in (0, 1, 2))
            env_idx = np.repeat(env, length)
            within = np.arange(int(length.sum())) - np.repeat(np.cumsum(length) - length, length)
            step_idx = (np.repeat(end - length, length) + within) % self._ring.shape[1]
            block = self._ring[t.from_numpy(env_idx).to(self.device), t.from_numpy(step_idx).to(self.device)].cpu().numpy()
            finished._add_packed(block, D, A, length)
        return finished

    def _collect_fused(self, n_moves):
        """``n_moves`` moves in launches of ``moves_per_launch``; what ended during a launch is read while the next one
        runs (the environments, their histories and the random streams live on the device: a launch needs nothing from
        the host, and the host reads finished episodes, not moves)."""
        bufs = self._fused_state()
        finished = EpisodeSeq()
        pending = []   # launches enqueued and not read yet, oldest first
        left, which = int(n_moves), 0
        while left > 0 or pending:
            # TWO launches are kept enqueued ahead of the one the host reads: the copy of a launch's arena is a blit kernel that
            # finds room on the CUs only as the NEXT launch drains, so the host gets launch k's records about when launch k + 1
            # ends -- with one launch ahead the GPU then idled while the host read them (2.4 of 7.3 ms per launch)
            while left > 0 and len(pending) < 2:
                k = min(left, self.moves_per_launch)
                self._launch_moves(k, bufs[which])
                pending.append(bufs[which])
                left -= k
                which = (which + 1) % len(bufs)
            finished.extend(self._episodes_of_launch(pending.pop(0)))
        return finished

    def collect(self, n_moves):
        """``n_moves`` moves of every environment -> the finished episodes (an EpisodeSeq: len / iteration / indexing)."""
        if self.fused_moves:
            out = self._collect_fused(n_moves)
            self.obs = self.env.observe()
        else:
            out = EpisodeSeq()
            for _ in range(n_moves):
                out.extend(self.play_move())
        self.tree.check()
        return out
