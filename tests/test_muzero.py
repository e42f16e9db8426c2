"""MuZero (BASELINE configs[4]).  The reference has no MuZero to pin against: the CPU
tests check the restatement of the published pseudocode / the Gymnasium CartPole equations against hand-computed
known answers and invariants; the GPU tests compare the HIP tree kernels with that restatement bit for bit
(the oracle is fed the very network outputs the device used) and the batched environment with the scalar one."""
import math

import numpy as np
import pytest

from oracle import muzero_ref as ref


# ------------------------------------------------------------------ oracle (CPU)
def test_cartpole_first_step_known_answer():
    """From rest at the origin, push right: the textbook numbers of the Barto-Sutton-Anderson cart-pole."""
    env = ref.RefCartPole()
    env.reset((0.0, 0.0, 0.0, 0.0))
    state, reward, terminated, truncated = env.step(1)
    temp = 10.0 / 1.1
    thetaacc = -temp / (0.5 * (4.0 / 3.0 - 0.1 / 1.1))
    xacc = temp - 0.05 * thetaacc / 1.1
    assert state == (0.0, 0.02 * xacc, 0.0, 0.02 * thetaacc)
    assert abs(state[1] - 0.19512195) < 1e-8 and abs(state[3] + 0.29268293) < 1e-8
    assert reward == 1.0 and not terminated and not truncated
    # constant push to the right: the pole falls left past 12 degrees within a dozen steps
    for n in range(2, 200):
        state, reward, terminated, truncated = env.step(1)
        if terminated:
            break
    assert terminated and 5 < n < 20 and state[2] < -ref.RefCartPole.theta_threshold_radians
    env.reset((0.0, 0.0, 0.0, 0.0))
    for n in range(1, 501):  # alternating pushes keep it up; the time limit truncates at 500
        state, reward, terminated, truncated = env.step(n % 2)
        if terminated or truncated:
            break
    assert n == 500 and truncated or terminated


def _toy_model(hidden, action, path):
    """A deterministic stand-in for the learned model: everything depends on the action path only."""
    h = hash(path) % 1000 / 1000.0
    p0 = 0.2 + 0.6 * ((len(path) * 7 + sum(path)) % 5) / 4.0
    return path, 1.0 if sum(path) % 3 else 0.5, (p0, 1.0 - p0), 10.0 * h - 3.0


def test_muzero_search_invariants_and_tie_break():
    cfg = ref.MuZeroConfig(num_simulations=50)
    root = ref.Node(0)
    ref.expand_node(root, (), 0.0, (0.5, 0.5))
    log = []
    stats = ref.run_mcts(cfg, root, _toy_model, log=log)
    assert log[0] == (1, )  # sqrt(0) makes every first score 0: max() over (score, action) takes the larger action
    assert root.visit_count == 50 and sum(c.visit_count for c in root.children) == 50
    dump = ref.tree_dump(root)
    assert sum(1 for p, (n, _, _, _) in dump.items() if n > 0) == 51  # root + one new node per simulation
    for path, (n, vsum, reward, prior) in dump.items():
        node = root
        for a in path:
            node = node.children[a]
        if node.expanded():
            assert node.visit_count == 1 + sum(c.visit_count for c in node.children) - (1 if not path else 0)
    assert stats.minimum <= root.value() <= stats.maximum
    # value recursion at the root: value_sum accumulates reward + discount * value along each path
    assert math.isfinite(root.value_sum) and stats.maximum > stats.minimum


def test_min_max_normalisation_matches_pseudocode():
    s = ref.MinMaxStats()
    assert s.normalize(3.0) == 3.0  # no range yet: value unchanged
    s.update(1.0)
    assert s.normalize(3.0) == 3.0  # max == min
    s.update(5.0)
    assert s.normalize(3.0) == 0.5 and s.normalize(1.0) == 0.0 and s.normalize(5.0) == 1.0


def test_initial_states_and_replay_targets():
    from rlzero_amd.muzero.cartpole import initial_states
    from rlzero_amd.muzero.selfplay import Episode, ReplayBuffer
    a = initial_states(3, np.arange(64), np.zeros(64, dtype=np.int64))
    b = initial_states(3, np.arange(32, 64), np.zeros(32, dtype=np.int64))
    assert a.shape == (64, 4) and (a >= -0.05).all() and (a < 0.05).all() and np.array_equal(a[32:], b)
    assert not np.array_equal(a, initial_states(3, np.arange(64), np.ones(64, dtype=np.int64)))
    ep = Episode()
    for t in range(6):
        ep.obs.append(np.full(4, t, dtype=np.float32))
        ep.actions.append(t % 2)
        ep.rewards.append(1.0)
        ep.policies.append(np.array([0.25, 0.75], dtype=np.float32))
        ep.root_values.append(float(10 + t))
    buf = ReplayBuffer(unroll_steps=2, td_steps=3, discount=0.5, seed=0)
    buf.add(ep)
    assert buf._value_target(ep, 0) == 1 + 0.5 + 0.25 + 13 * 0.125      # 3 rewards + bootstrap from root value 3
    assert buf._value_target(ep, 4) == 1 + 0.5                           # runs off the end: no bootstrap
    obs, actions, tv, tr, tp, mask = buf.sample(16, 2)
    assert obs.shape == (16, 4) and actions.shape == (16, 2) and tv.shape == tr.shape == mask.shape == (16, 3)
    for b_ in range(16):
        t = int(obs[b_, 0])
        for k in range(3):
            inside = t + k < 6
            assert mask[b_, k] == (1.0 if inside else 0.0)
            assert tv[b_, k] == (np.float32(buf._value_target(ep, t + k)) if inside else 0.0)
            if k > 0:
                assert tr[b_, k] == (1.0 if t + k - 1 < 6 else 0.0)
                if t + k - 1 < 6:
                    assert actions[b_, k - 1] == (t + k - 1) % 2


# ------------------------------------------------------------------ HIP tree kernels vs the oracle
def _fake_selfplay(n_envs, hist=64):
    """The host-side state MuZeroSelfPlay._ingest works on, without a GPU."""
    import types
    fake = types.SimpleNamespace(net=types.SimpleNamespace(obs_dim=4), n_actions=2, n_envs=n_envs, HIST=hist, _t=0, moves_done=0)
    fake._h_obs = np.zeros((hist, n_envs, 4), dtype=np.float32)
    fake._h_act = np.zeros((hist, n_envs), dtype=np.int64)
    fake._h_rew = np.zeros((hist, n_envs), dtype=np.float64)
    fake._h_pol = np.zeros((hist, n_envs, 2), dtype=np.float32)
    fake._h_val = np.zeros((hist, n_envs), dtype=np.float64)
    fake._ep_start = np.zeros(n_envs, dtype=np.int64)
    return fake


def test_ingest_of_a_stretch_of_moves_equals_move_by_move():
    """The records of K moves ingested in one call (what the fused moves hand over) against the same records one move
    at a time: same episodes in the same order -- including environments that finish twice inside the stretch and
    episodes that straddle calls and the wrap of the history ring."""
    from rlzero_amd.muzero.selfplay import MuZeroSelfPlay
    rng = np.random.RandomState(3)
    n_envs, n_moves = 13, 90
    rec = np.zeros((n_moves, n_envs, 10))
    rec[:, :, :4] = rng.randn(n_moves, n_envs, 4).astype(np.float32)
    rec[:, :, 4] = rng.randint(2, size=(n_moves, n_envs))
    rec[:, :, 5] = 1.0
    rec[:, :, 6:8] = rng.randint(1, 20, size=(n_moves, n_envs, 2))
    rec[:, :, 8] = rng.randn(n_moves, n_envs)
    rec[:, :, 9] = rng.rand(n_moves, n_envs) < 0.3          # short episodes: several ends per environment and stretch
    one, many = _fake_selfplay(n_envs), _fake_selfplay(n_envs)
    a = []
    for t in range(n_moves):
        a.extend(MuZeroSelfPlay._ingest(one, rec[t]))
    b = []
    for lo_, hi_ in ((0, 7), (7, 8), (8, 24), (24, 40), (40, 90)):
        chunk = MuZeroSelfPlay._ingest(many, rec[lo_:hi_])
        assert len(chunk) == int(rec[lo_:hi_, :, 9].sum()) == len(chunk.lengths())
        b.extend(chunk)
    assert len(a) == len(b) == int(rec[:, :, 9].sum()) and one._t == many._t == n_moves
    assert np.array_equal(one._ep_start, many._ep_start) and one.moves_done == many.moves_done == n_envs * n_moves
    for x, y in zip(a, b):
        assert len(x) == len(y) > 0
        for f in ('obs', 'actions', 'rewards', 'policies', 'root_values'):
            assert np.array_equal(getattr(x, f), getattr(y, f)), f
    first = a[0]   # and an episode is what the records say: the first one ends at the first done flag (move-major order)
    k0, e0 = np.argwhere(rec[:, :, 9] != 0)[0]
    assert len(first) == k0 + 1 and np.array_equal(first.obs, rec[:k0 + 1, e0, :4].astype(np.float32))
    assert np.array_equal(first.actions, rec[:k0 + 1, e0, 4].astype(np.int64))


def test_episode_sequence_is_a_lazy_view_of_the_finished_episodes():
    """EpisodeSeq (what MuZeroSelfPlay.collect returns): length, iteration, indexing (negative, slices), lengths() and
    extend() over chunks whose episodes lie in arbitrary order inside one flat block (the device arena's layout)."""
    from rlzero_amd.muzero.selfplay import EpisodeSeq
    block = np.arange(40, dtype=np.float64)
    fields = (np.stack([block] * 4, axis=1).astype(np.float32), block.astype(np.int64), block * 0 + 1.0,
              np.stack([block, block], axis=1).astype(np.float32), -block)
    seq = EpisodeSeq()
    assert len(seq) == 0 and list(seq) == [] and not seq and len(seq.lengths()) == 0
    seq._add(fields, lengths=[3, 5, 2], first_rows=[10, 0, 30])      # three episodes out of order inside the block
    other = EpisodeSeq()
    other._add(tuple(f[20:26] for f in fields), lengths=[4, 2])      # contiguous episodes: rows 0-3, 4-5 of this block
    seq.extend(other)
    assert len(seq) == 5 and bool(seq) and list(seq.lengths()) == [3, 5, 2, 4, 2]
    want_first = [10, 0, 30, 20, 24]
    for i, ep in enumerate(seq):
        assert len(ep) == seq.lengths()[i] and ep.actions[0] == want_first[i] and ep.obs.shape == (len(ep), 4)
        assert np.array_equal(ep.root_values, -ep.actions.astype(np.float64)) and np.array_equal(seq[i].actions, ep.actions)
    assert seq[-1].actions.tolist() == [24, 25] and [len(e) for e in seq[1:4]] == [5, 2, 4]
    with pytest.raises(IndexError):
        seq[5]


def _hexf(x):
    return float(x).hex()


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [True, False])
def test_muzero_tree_kernels_bit_exact_vs_pseudocode(fused):
    """Both routes -- the whole search in one launch (k_mz_search, the model evaluated inside the kernel) and the
    step-by-step route (tree kernels + PyTorch layers) -- against the CPython restatement of the pseudocode fed the network
    outputs the device used: identical trees, statistic for statistic."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(1)
    net = MuZeroNet().to('cuda:0').eval()
    env = CartPoleBatch(48, 'cuda:0', seed=2)
    sp = MuZeroSelfPlay(net, env, n_sims=50, seed=5, fused=fused)
    assert sp.fused == fused
    for trial in range(2):
        record = []
        visits, root_value = sp.search(env.observe(), add_noise=True, record=record)
        tree = sp.tree
        n, vsum, vmin, vmax = (x.cpu().numpy().copy() for x in tree.root_stats())
        child_sum = tree.root_children('value_sum').cpu().numpy()
        child_rew = tree.root_children('reward').cpu().numpy()
        child_pri = tree.root_children('prior').cpu().numpy()
        visits = visits.cpu().numpy()
        _, probs0, noise = record[0]
        sims = [tuple(x.cpu().numpy() for x in r) for r in record[1:]]
        probs0, noise = probs0.cpu().numpy(), noise.cpu().numpy()
        cfg = ref.MuZeroConfig(num_simulations=50)
        for g in range(0, 48, 5):
            outputs = {}
            root = ref.Node(0)
            ref.expand_node(root, None, 0.0, [float(p) for p in probs0[g]])
            ref.add_exploration_noise(cfg, root, [float(x) for x in noise[g]])
            step = [0]

            def model(hidden, action, path, g=g, step=step):
                parent, act, leaf, reward, probs, value = sims[step[0]]
                assert act[g] == action  # the device took the same edge in the same simulation
                step[0] += 1
                return None, float(reward[g]), [float(p) for p in probs[g]], float(value[g])

            stats = ref.run_mcts(cfg, root, model)
            assert root.visit_count == n[g] == 50 and _hexf(root.value_sum) == _hexf(vsum[g])
            assert _hexf(stats.minimum) == _hexf(vmin[g]) and _hexf(stats.maximum) == _hexf(vmax[g])
            for a, child in enumerate(root.children):
                assert child.visit_count == visits[g, a]
                assert _hexf(child.value_sum) == _hexf(child_sum[g, a])
                assert _hexf(child.reward) == _hexf(child_rew[g, a]) and _hexf(child.prior) == _hexf(child_pri[g, a])
            assert _hexf(root.value()) == _hexf(root_value[g].item())
        env.step(torch.from_numpy(visits.argmax(axis=1)).to('cuda:0'))
    # the production path (fused: no trace arrays; else the hipGraph replay of the simulation step) gives the traced path's trees
    obs = env.observe()
    v_graph, rv_graph = sp.search(obs, add_noise=False)
    v_graph, rv_graph = v_graph.clone(), rv_graph.clone()
    v_eager, rv_eager = sp.search(obs, add_noise=False, record=[])
    assert (sp._graph is not None) != fused and torch.equal(v_graph, v_eager) and torch.equal(rv_graph, rv_eager)
    sp.tree.check()
    sp.close()


class _BareEnv(object):
    """What MuZeroSelfPlay.search needs of an environment (no stepping): sizes, device, observations."""
    max_episode_steps = 500

    def __init__(self, n_envs, n_actions, obs):
        self.n_envs, self.n_actions, self.device, self._obs = n_envs, n_actions, obs.device, obs

    def observe(self):
        return self._obs


@pytest.mark.gpu
@pytest.mark.parametrize('n_actions,n_sims', [(1, 12), (3, 30), (4, 25), (6, 10), (6, 50), (8, 40)])
def test_fused_search_other_action_counts_and_tree_placements(n_actions, n_sims):
    """k_mz_search away from CartPole's two actions: 3 / 4 actions take the quad's second exchange, more than 4 the
    strided scan, and (6, 50) / (8, 40) trees do not fit the LDS budget and stay in HBM (one-lane walk) -- every one
    against the CPython restatement fed the kernel's traced network outputs: identical visit counts and value sums."""
    import torch
    from rlzero_amd.muzero import MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(10 + n_actions)
    G = 37
    net = MuZeroNet(n_actions=n_actions).to('cuda:0').eval()
    obs = torch.randn(G, 4, device='cuda:0')
    sp = MuZeroSelfPlay(net, _BareEnv(G, n_actions, obs), n_sims=n_sims, seed=3, fused=True)
    assert sp.fused and not sp.fused_moves
    record = []
    visits, root_value = sp.search(obs, add_noise=True, record=record)
    visits, root_value = visits.cpu().numpy(), root_value.cpu().numpy()
    n, vsum, vmin, vmax = (x.cpu().numpy().copy() for x in sp.tree.root_stats())
    child_sum = sp.tree.root_children('value_sum').cpu().numpy()
    _, probs0, noise = record[0]
    probs0, noise = probs0.cpu().numpy(), noise.cpu().numpy()
    sims = [tuple(x.cpu().numpy() for x in r) for r in record[1:]]
    cfg = ref.MuZeroConfig(num_simulations=n_sims)
    for g in range(G):
        root = ref.Node(0)
        ref.expand_node(root, None, 0.0, [float(p) for p in probs0[g]])
        ref.add_exploration_noise(cfg, root, [float(x) for x in noise[g]])
        step = [0]

        def model(hidden, action, path, g=g, step=step):
            parent, act, leaf, reward, probs, value = sims[step[0]]
            assert act[g] == action
            step[0] += 1
            return None, float(reward[g]), [float(p) for p in probs[g]], float(value[g])

        stats = ref.run_mcts(cfg, root, model)
        assert root.visit_count == n[g] == n_sims and _hexf(root.value_sum) == _hexf(vsum[g])
        assert _hexf(stats.minimum) == _hexf(vmin[g]) and _hexf(stats.maximum) == _hexf(vmax[g])
        for a, child in enumerate(root.children):
            assert child.visit_count == visits[g, a] and _hexf(child.value_sum) == _hexf(child_sum[g, a])
    # and the untraced launch builds the same trees
    v2, rv2 = sp.search(obs, add_noise=False)
    v2, rv2 = v2.clone(), rv2.clone()
    v3, rv3 = sp.search(obs, add_noise=False, record=[])
    assert torch.equal(v2, v3) and torch.equal(rv2, rv3)
    sp.tree.check()
    sp.close()


@pytest.mark.gpu
def test_fused_search_does_not_depend_on_how_games_share_workgroups():
    """A game's search is a column of the workgroup's tiles and a quad of its wave: 4, 8 or 16 games per workgroup
    (rz_mz_set_search_shape) give bit-identical visit counts, root values and hidden states."""
    import torch
    from rlzero_amd.muzero import MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(21)
    G = 53
    net = MuZeroNet().to('cuda:0').eval()
    obs = torch.randn(G, 4, device='cuda:0')
    sp = MuZeroSelfPlay(net, _BareEnv(G, 2, obs), n_sims=40, seed=2, fused=True)
    got = []
    for gpw in (16, 8, 4, 0):
        sp.tree.set_search_shape(gpw)
        visits, root_value = sp.search(obs, add_noise=False)
        got.append((visits.clone(), root_value.clone(), sp.hidden.clone(), sp.tree.root_children('value_sum')))
    for other in got[1:]:
        for a, b in zip(got[0], other):
            assert torch.equal(a, b)
    sp.close()


@pytest.mark.gpu
def test_fused_moves_history_ring_wraps():
    """More moves than the device ring has steps: episodes that straddle the wrap (and launches of uneven length) still come
    out as the scalar environment replays them under their recorded actions."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    from rlzero_amd.muzero.cartpole import initial_states
    torch.manual_seed(13)
    net = MuZeroNet().to('cuda:0').eval()
    G = 20
    env = CartPoleBatch(G, 'cuda:0', seed=17)
    sp = MuZeroSelfPlay(net, env, n_sims=4, seed=1, moves_per_launch=13)
    episodes = []
    for n in (300, 1, 290, 64):     # 655 moves > 500 + 2 * 13 + 12 ring steps
        episodes.extend(sp.collect(n))
    ring, ep_start = sp.device_history()
    assert sp._t == 655 > ring.shape[1] and len(episodes) > 200
    count = np.zeros(G, dtype=np.int64)
    for ep in episodes:  # which environment an episode belongs to: its first observation is that environment's next start
        first = np.asarray(ep.obs[0], dtype=np.float32)
        owner = [i for i in range(G) if np.array_equal(initial_states(17, [i], [count[i]])[0].astype(np.float32), first)]
        assert len(owner) == 1
        i = owner[0]
        r = ref.RefCartPole()
        r.reset(initial_states(17, [i], [count[i]])[0])
        for t in range(len(ep)):
            assert np.max(np.abs(np.asarray(ep.obs[t]) - np.array(r.state, dtype=np.float64).astype(np.float32))) < 1e-6
            state, rew, term, trunc = r.step(int(ep.actions[t]))
            assert (term or trunc) == (t == len(ep) - 1)
        count[i] += 1
    assert np.array_equal(count, env.episode) and sum(len(ep) for ep in episodes) == int(ep_start.sum())
    sp.close()


@pytest.mark.gpu
@pytest.mark.parametrize('temperature', [1.0, 0.5])
def test_fused_moves_draw_actions_from_the_visit_counts(temperature):
    """select_action of the pseudocode inside the kernel: P(action) = visits ^ (1 / T) / sum, drawn from the kernel's
    counter-based stream -- over 8192 environments the drawn actions follow the recorded visit counts (mean of the
    per-environment probability of action 1, 4 sigma), and every drawn action has a visit."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(12)
    net = MuZeroNet().to('cuda:0').eval()
    G = 8192
    sp = MuZeroSelfPlay(net, CartPoleBatch(G, 'cuda:0', seed=31), n_sims=16, seed=6, temperature=temperature, moves_per_launch=2)
    assert sp.fused_moves
    sp.collect(2)
    ring, _ = sp.device_history()
    for t in range(2):
        action, visits = ring[:, t, 4].astype(np.int64), ring[:, t, 6:8]
        assert (visits.sum(axis=1) == 16).all() and (visits[np.arange(G), action] > 0).all()
        w = visits ** (1.0 / temperature)
        p1 = w[:, 1] / w.sum(axis=1)
        sigma = math.sqrt(float((p1 * (1 - p1)).sum())) / G
        assert abs(float(action.mean()) - float(p1.mean())) < 4 * sigma + 1e-9, (action.mean(), p1.mean(), sigma)
    a0, a1 = ring[:, 0, 4], ring[:, 1, 4]
    assert 0.2 < float((a0 != a1).mean()) < 0.8   # the two moves of an environment are separate draws
    sp.close()


@pytest.mark.gpu
def test_fused_moves_episodes_that_do_not_fit_the_arena_come_from_the_ring():
    """A per-launch arena far too small for what ends during a launch: the kernel marks those episodes (row -1) and the
    host reads them back from the device ring -- the same episodes as with the default arena."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(7)
    net = MuZeroNet().to('cuda:0').eval()

    def play(arena_rows):
        sp = MuZeroSelfPlay(net, CartPoleBatch(96, 'cuda:0', seed=5), n_sims=10, seed=2, moves_per_launch=9, arena_rows=arena_rows)
        eps = sp.collect(45)
        out = sorted((len(e), e.obs.tobytes(), e.actions.tobytes(), e.root_values.tobytes(), e.policies.tobytes()) for e in eps)
        sp.close()
        return out

    roomy, tight = play(None), play(40)
    assert len(roomy) > 100 and roomy == tight


@pytest.mark.gpu
def test_fused_search_network_matches_the_torch_model():
    """The recurrent inference k_mz_search evaluates inside the kernel (dynamics, reward head, min-max scaled next state,
    prediction with softmax) against MuZeroNet.recurrent_inference on the same (parent state, action), simulation by
    simulation: 1e-5 on reward / value / probabilities and on the hidden state stored for the leaf; ragged batch (not a
    multiple of 64 games), weights re-uploaded after they change."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(4)
    net = MuZeroNet().to('cuda:0').eval()
    env = CartPoleBatch(100, 'cuda:0', seed=3)
    sp = MuZeroSelfPlay(net, env, n_sims=30, seed=1)
    assert sp.fused
    for round_ in range(2):
        record = []
        sp.search(env.observe(), add_noise=True, record=record)
        hidden = sp.hidden.clone()
        rows = torch.arange(100, device='cuda:0')
        worst = 0.0
        with torch.no_grad():
            for parent, action, leaf, reward, probs, value in record[1:]:
                nxt, r, logits, v = net.recurrent_inference(hidden[rows, parent.long()], action.long())
                worst = max(worst, float((r - reward).abs().max()), float((v - value).abs().max()),
                            float((torch.softmax(logits, dim=1) - probs).abs().max()),
                            float((nxt - hidden[rows, leaf.long()]).abs().max()))
        assert worst <= 1e-5, worst
        n, _, _, _ = sp.tree.root_stats()
        assert (n == 30).all()
        with torch.no_grad():  # a "learner step": the next search must see the new weights
            for p in net.parameters():
                p.add_(0.01 * torch.randn_like(p))
    sp.tree.check()
    sp.close()


@pytest.mark.gpu
def test_cartpole_batch_vs_scalar_restatement():
    import torch
    from rlzero_amd.muzero.cartpole import CartPoleBatch, initial_states
    env = CartPoleBatch(32, 'cuda:0', seed=9)
    init = initial_states(9, np.arange(32), np.zeros(32, dtype=np.int64))
    assert np.array_equal(env.state.cpu().numpy(), init)
    refs = [ref.RefCartPole() for _ in range(32)]
    for r, s in zip(refs, init):
        r.reset(s)
    rng = np.random.RandomState(0)
    alive = np.ones(32, dtype=bool)
    for t in range(120):
        actions = rng.randint(2, size=32)
        before = env.state.cpu().numpy().copy()
        obs, reward, terminated, truncated = env.step(torch.from_numpy(actions).to('cuda:0'))
        terminated = terminated.cpu().numpy()
        for i in range(32):
            if not alive[i]:
                continue
            state, rew, term, trunc = refs[i].step(int(actions[i]))
            assert term == bool(terminated[i])
            if term:
                alive[i] = False  # the batch auto-resets; the scalar twin stops here
                assert np.array_equal(env.state[i].cpu().numpy(), initial_states(9, [i], [1])[0])
            else:
                assert np.max(np.abs(env.state[i].cpu().numpy() - np.array(state))) < 1e-9
        assert (reward.cpu().numpy() == 1.0).all() and before.shape == (32, 4)
    assert (~alive).sum() > 10  # random play drops the pole quickly


@pytest.mark.gpu
@pytest.mark.parametrize('fused_moves', [True, False])
def test_muzero_selfplay_paths_agree_on_what_a_move_is(fused_moves):
    """Whole moves inside the kernel (rz_mz_play_cartpole) against the host-driven loop: the records of both follow the
    scalar CartPole restatement under the recorded actions (observations, resets, done flags), every search spends its
    simulations, and at temperature 0 the action is the arg-max of the visit counts."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    from rlzero_amd.muzero.cartpole import initial_states
    torch.manual_seed(5)
    net = MuZeroNet().to('cuda:0').eval()
    G, n_moves = 40, 30
    env = CartPoleBatch(G, 'cuda:0', seed=11)
    sp = MuZeroSelfPlay(net, env, n_sims=25, seed=4, temperature=0.0, fused_moves=fused_moves, moves_per_launch=7)
    assert sp.fused and sp.fused_moves == fused_moves
    episodes = sp.collect(n_moves)
    assert sp.sims_done == G * n_moves * 25 and sp.moves_done == G * n_moves and sp._t == n_moves
    if fused_moves:  # the history is on the device: [environment, step % ring_steps, obs | action | reward | visits | value | done]
        ring, ep_start = sp.device_history()

        def record(t, i):
            r = ring[i, t % ring.shape[1]]
            return r[:4].astype(np.float32), int(r[4]), r[5], (r[6:8] / r[6:8].sum()).astype(np.float32), r[8], bool(r[9])
    else:
        ep_start = sp._ep_start

        def record(t, i):
            slot = t % sp.HIST
            return sp._h_obs[slot, i], int(sp._h_act[slot, i]), sp._h_rew[slot, i], sp._h_pol[slot, i], sp._h_val[slot, i], None
    refs, episode = [], np.zeros(G, dtype=np.int64)
    for i in range(G):
        r = ref.RefCartPole()
        r.reset(initial_states(11, [i], [0])[0])
        refs.append(r)
    ended, want_episodes = 0, []
    running = [[] for _ in range(G)]
    for t in range(n_moves):
        for i in range(G):
            obs, action, reward, pol, value, done = record(t, i)
            want_obs = np.array(refs[i].state, dtype=np.float64).astype(np.float32)
            assert np.max(np.abs(obs - want_obs)) < 1e-6, (t, i)
            assert abs(pol.sum() - 1.0) < 1e-6 and np.all(np.round(pol * 25) == pol * 25)   # visit counts / 25
            assert pol[action] == pol.max()                                                  # temperature 0
            assert reward == 1.0 and math.isfinite(value)
            running[i].append((obs, action, value))
            state, rew, term, trunc = refs[i].step(action)
            assert done is None or done == (term or trunc)
            if term or trunc:
                ended += 1
                episode[i] += 1
                refs[i] = ref.RefCartPole()
                refs[i].reset(initial_states(11, [i], [episode[i]])[0])
                want_episodes.append(running[i])
                running[i] = []
    assert ended == len(episodes) > 0 and np.array_equal(env.episode, episode)
    # the episodes handed out: ordered by (last move, environment), each the run of its environment's records
    for ep, want in zip(episodes, want_episodes):
        assert len(ep) == len(want)
        assert np.array_equal(ep.obs, np.array([w[0] for w in want])) and list(ep.actions) == [w[1] for w in want]
        assert np.array_equal(ep.root_values, np.array([w[2] for w in want])) and (np.asarray(ep.rewards) == 1.0).all()
    final = env.state.cpu().numpy()
    for i in range(G):
        assert np.max(np.abs(final[i] - np.array(refs[i].state))) < 1e-9
    assert sum(len(ep) for ep in episodes) == int(sum(ep_start))
    sp.tree.check()
    sp.close()


@pytest.mark.gpu
def test_fused_moves_search_agrees_with_the_traced_search_and_noise_is_dirichlet():
    """(a) noise off: the searches inside rz_mz_play_cartpole start from the kernel's own initial inference (matrix
    pipe) instead of torch's, so visit counts agree with MuZeroSelfPlay.search on the same observations except where a
    last-bit difference flips a near-tie; (b) noise weight 1: the root priors ARE the kernel's Dirichlet(alpha) draw:
    mean 1/2, variance 1 / (4 (2 alpha + 1))."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(6)
    net = MuZeroNet().to('cuda:0').eval()
    G = 512
    env = CartPoleBatch(G, 'cuda:0', seed=21)
    sp = MuZeroSelfPlay(net, env, n_sims=30, seed=8, root_exploration_fraction=0.0, fused_moves=True, moves_per_launch=1)
    obs = env.observe().clone()
    visits_host, value_host = sp.search(obs, add_noise=False)   # torch initial inference + fused search
    visits_host, value_host = visits_host.cpu().numpy(), value_host.cpu().numpy()
    sp.collect(1)
    ring, _ = sp.device_history()
    visits_dev = ring[:, 0, 6:8].astype(np.int64)
    assert (visits_dev.sum(axis=1) == 30).all()
    same = (visits_dev == visits_host).all(axis=1)
    assert same.mean() >= 0.9, same.mean()
    assert np.max(np.abs(ring[:, 0, 8][same] - value_host[same])) < 1e-4
    sp.close()
    for alpha in (0.25, 1.5):
        env = CartPoleBatch(4096, 'cuda:0', seed=22)
        sp = MuZeroSelfPlay(net, env, n_sims=2, seed=9, root_dirichlet_alpha=alpha, root_exploration_fraction=1.0,
                            fused_moves=True, moves_per_launch=1)
        sp.collect(1)
        pri = sp.tree.root_children('prior').cpu().numpy()
        assert np.max(np.abs(pri.sum(axis=1) - 1.0)) < 1e-6 and (pri >= 0).all()
        want_var = 1.0 / (4.0 * (2.0 * alpha + 1.0))
        assert abs(pri[:, 0].mean() - 0.5) < 0.03 and abs(pri[:, 0].var() - want_var) < 0.1 * want_var + 0.004, (alpha, pri[:, 0].var())
        sp.close()


@pytest.mark.gpu
def test_muzero_selfplay_and_learner_smoke():
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroAgent, MuZeroSelfPlay, ReplayBuffer
    torch.manual_seed(0)
    agent = MuZeroAgent(device='cuda:0')
    env = CartPoleBatch(64, 'cuda:0', seed=1)
    sp = MuZeroSelfPlay(agent.net, env, n_sims=20, seed=3)
    buf = ReplayBuffer(unroll_steps=5, td_steps=10, seed=0)
    for ep in sp.collect(40):
        buf.add(ep)
    assert len(buf) > 20 and sp.sims_done == 64 * 40 * 20
    lengths = [len(ep) for ep in buf.episodes]
    assert min(lengths) >= 7 and max(lengths) <= 500
    first = None
    for it in range(30):
        loss, lv, lr_, lp = agent.learn(buf.sample(128, 2))
        assert math.isfinite(loss)
        first = first if first is not None else loss
    assert loss < first  # the model fits its own replay data
    sp.close()


@pytest.mark.gpu
def test_whole_moves_network_on_the_f16_pipe_agrees_with_the_f32_search():
    """Whole MOVES run the four 64 x 64 layers of a simulation on the f16 matrix pipe (hi + lo operand pairs, hidden states
    kept as f16 pieces); the search of a move launched from the host runs them on the f32-input MFMA.  From the same
    observations, without root noise, both must arrive at the same search: root values within 1e-4, visit counts equal in
    (nearly) every environment (a 1e-7 difference of a network output may flip a tie); also on weights 20x the initial scale."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    for gain in (1.0, 20.0):
        torch.manual_seed(7)
        net = MuZeroNet().to('cuda:0').eval()
        with torch.no_grad():
            for name, p_ in net.named_parameters():
                if gain != 1.0 and name.split('.')[0] in ('dyn1', 'pre1') and name.endswith('weight'):
                    p_.mul_(gain)
        G = 200
        out = {}
        for moves in (True, False):
            sp = MuZeroSelfPlay(net, CartPoleBatch(G, 'cuda:0', seed=3), n_sims=50, seed=9, temperature=0.0,
                                root_exploration_fraction=0.0, fused=True, fused_moves=moves, moves_per_launch=1)
            sp.collect(1)
            if moves:
                ring, _ = sp.device_history()
                r = ring[:, 0]
                out[moves] = (r[:, 6:8].astype(np.int64), r[:, 8].copy(), r[:, 4].astype(np.int64))
            else:
                out[moves] = (np.round(sp._h_pol[0] * 50).astype(np.int64), sp._h_val[0].copy(), sp._h_act[0].copy())
            sp.tree.check()
            sp.close()
        same = (out[True][0] == out[False][0]).all(axis=1)
        assert same.mean() >= 0.97, (gain, float(same.mean()))
        assert np.max(np.abs(out[True][1][same] - out[False][1][same])) <= 1e-4, gain
        assert (out[True][2][same] == out[False][2][same]).all()


@pytest.mark.gpu
def test_whole_moves_refuse_weights_without_finite_bounds():
    """The f16 layers of whole MOVES have no overflow path for finite weights (bounds in rz_mz_load_model); a model with inf / nan
    weights gives no bound, and the whole-moves route then fails loudly instead of filling the trees with inf: the move-by-move
    search still runs."""
    import torch
    from rlzero_amd._hip import HipError
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    torch.manual_seed(0)
    net = MuZeroNet().to('cuda:0').eval()
    with torch.no_grad():
        net.dyn2.weight.view(-1)[3] = float('inf')
    sp = MuZeroSelfPlay(net, CartPoleBatch(32, 'cuda:0', seed=0), n_sims=8, seed=0, fused=True)
    assert sp.fused_moves
    with pytest.raises(HipError, match='non-finite weights'):
        sp.collect(2)
    sp.close()
