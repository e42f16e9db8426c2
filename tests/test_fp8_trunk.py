"""RZ_NET_SPLIT_F16_FP8 (include/rlzero_hip.h), the OPT-IN trunk whose conv3 cross terms run on the block-scaled FP8 pipe: narrower
arithmetic than the reference's f32 (rlzero/games/gomoku/policy_value_net.py:34-52), so nothing here is a parity claim.  Pinned:
(1) the device computes what the mode's float64 model computes (oracle/fp8_cross_ref.py) -- layout, scales, conversions --, far
closer than the mode's own distance from the float64 forward, and that distance is printed and bounded; (2) the mode is refused
where it does not exist (float planes, boards outside 11 .. 16); (3) the resident search and the two-launch step agree bit for bit
in this mode too."""
import numpy as np
import pytest

from oracle import fp8_cross_ref as model
from oracle.gomoku_ref import RefGomoku

pytestmark = pytest.mark.gpu


def _positions(B, n, count, seed):
    rs = np.random.RandomState(seed)
    envs = [RefGomoku.from_moves(B, n, [0])]
    while len(envs) < count:
        e = RefGomoku(B, n)
        for m in rs.permutation(B * B)[:rs.randint(1, B * B - 1)]:
            e.step(int(m))
            if e.game_end_winner()[0]:
                break
        if not e.game_end_winner()[0]:
            envs.append(e)
    return envs


def _set_roots(eng, envs):
    from rlzero_amd.engine import int_to_bits
    stones = np.array([[int_to_bits(e.bitboards()[0]), int_to_bits(e.bitboards()[1])] for e in envs], dtype=np.uint64)
    eng.set_roots(stones, [e.current_player() for e in envs], [e.last_move for e in envs], reset_trees=True)


def _net(B, seed, gain):
    """PolicyValueNet(B) at PyTorch's default init with the trunk's and the policy layer's weights times ``gain``: gain 1 is the
    benchmark's net (log-probabilities within 0.1 of each other), gain 3 a policy as sharp as a trained one (spread ~10)."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(seed)
    net = PolicyValueNet(B)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.startswith('conv') or name == 'act_fc1.weight':
                p.mul_(gain)
    return net


def _expand_roots(net, B, envs, algo, deferred):
    """One simulation per game (the root's expansion) -> (log priors over each game's legal moves, root values)."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=len(envs))
    evaluator.hip.set_algo(algo)
    evaluator.deferred_priors = deferred
    evaluator.resident_search = False
    eng = MCTSEngine(B, 5, n_games=len(envs), n_playout=4, device='cuda:0')
    _set_roots(eng, envs)
    eng.sim_chunk(evaluator, 1)
    pri = eng.root_priors().astype(np.float64)
    _, rw = eng.root_stats()
    eng.check()
    eng.close()
    evaluator.hip.close()
    return pri, -np.asarray(rw, dtype=np.float64)


@pytest.mark.parametrize('B', [15, 11, 16])
def test_fp8_cross_terms_compute_their_model_and_say_their_error(B):
    import torch
    envs = _positions(B, 5, 12, seed=B)
    planes = torch.from_numpy(np.stack([e.current_state() for e in envs]).astype(np.float32))
    report = []
    for gain in (1.0, 3.0):
        net = _net(B, seed=100 + B, gain=gain)
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        want = {mode: model.forward(sd, planes, mode) for mode in ('f64', 'split', 'fp8', 'f16')}
        legal = [np.array(e.leagel_actions()) for e in envs]

        def dist(logp_a, logp_b):
            return max(float(np.max(np.abs(np.asarray(logp_a[g])[legal[g]] - np.asarray(logp_b[g])[legal[g]]))) for g in range(len(envs)))
        got = {}
        for algo in ('split_f16', 'split_f16_fp8'):
            for deferred in (False, True):
                pri, val = _expand_roots(net, B, envs, algo, deferred)
                with np.errstate(divide='ignore'):
                    got[algo, deferred] = (np.log(pri), val)
        f64 = want['f64'][0].numpy()
        err_default = dist(got['split_f16', False][0], f64)
        err_fp8 = dist(got['split_f16_fp8', False][0], f64)
        err_f16_model = dist(want['f16'][0].numpy(), f64)
        err_fp8_model = dist(want['fp8'][0].numpy(), f64)
        off_model = dist(got['split_f16_fp8', False][0], want['fp8'][0].numpy())
        spread = float((want['f64'][0].max(1).values - want['f64'][0].min(1).values).mean())
        report.append('B %d gain %g (log-prob spread %.2f): |dlogp| vs float64: default %.2e, FP8 cross terms %.2e (model %.2e), plain f16 '
                      'would give %.2e; device vs its model %.2e' % (B, gain, spread, err_default, err_fp8, err_fp8_model, err_f16_model, off_model))
        # the mode is ON (its results differ from the default's) and computes its model, not something near it: a wrong byte order
        # or scale would leave the error at the plain-f16 level or far above
        assert not (np.array_equal(got['split_f16_fp8', False][0], got['split_f16', False][0]) and
                    np.array_equal(got['split_f16_fp8', False][1], got['split_f16', False][1]))
        # 3e-6: the f32 heads and softmax of both, at gain 1 the larger part.  0.4 of the mode's own error: the pipe sums a K = 128
        # block's products on a grid ~2^-13 below the block's largest product (profiles/microbench/fp8_scaled_mfma_check.hip),
        # which the model (exact sums) does not do -- measured 0.15 .. 0.22
        assert off_model <= max(0.4 * err_fp8_model, 3e-6), report[-1]
        if gain > 1.0:
            assert err_fp8 > 4.0 * err_default, report[-1]
        assert err_fp8 <= max(0.25 * err_f16_model, 3e-6), report[-1]
        assert err_default <= 1e-4 * max(1.0, gain)
        if gain == 1.0:   # the benchmark's net: inside the 1e-4 of the path with a wide margin; sharper policies are NOT (printed)
            assert err_fp8 <= 5e-6, report[-1]
        # both routes of the mode (priors inside the tree step / deferred) hand out the same priors, values to f32 rounding
        assert np.array_equal(got['split_f16_fp8', False][0], got['split_f16_fp8', True][0])
        assert np.max(np.abs(got['split_f16_fp8', False][1] - got['split_f16_fp8', True][1])) <= 2e-6
        assert np.max(np.abs(got['split_f16_fp8', False][1] - want['fp8'][1].numpy())) <= max(2e-6, 0.2 * float((want['fp8'][1] - want['f64'][1]).abs().max()) + 1e-6)
    print('\n' + '\n'.join(report))


def test_fp8_mode_is_refused_where_it_does_not_exist():
    import torch
    from rlzero_amd.engine import HipError, HipNet
    small = HipNet(9, 'cuda:0', max_boards=4)
    with pytest.raises(HipError):
        small.set_algo('split_f16_fp8')
    small.close()
    net = _net(15, seed=1, gain=1.0)
    hip = HipNet(15, 'cuda:0', max_boards=4)
    hip.load_state_dict(net.state_dict())
    obs = torch.zeros(2, 4, 15, 15, device='cuda:0')
    hip.forward(obs)
    hip.set_algo('split_f16_fp8')
    with pytest.raises(HipError):   # float planes would have to run another arithmetic: refused, not substituted
        hip.forward(obs)
    with pytest.raises(HipError):
        hip.trunk(obs)
    hip.set_algo('split_f16')
    hip.forward(obs)
    hip.close()


def test_fp8_resident_search_is_the_two_launch_step():
    """The one-launch search (rz_net_search_resident) and the two-launch step share the mode's arithmetic: same trees, bit for bit."""
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
    B, sims = 15, 60
    net = _net(B, seed=7, gain=2.0)
    envs = _positions(B, 5, 6, seed=3)
    dumps = {}
    for resident in (True, False):
        evaluator = HipNetEvaluator(net, B, 'cuda:0', max_boards=len(envs))
        evaluator.hip.set_algo('split_f16_fp8')
        evaluator.resident_search = resident
        eng = MCTSEngine(B, 5, n_games=len(envs), n_playout=sims, device='cuda:0', add_noise=True, noise_seed=5)
        assert evaluator.resident_ok(eng) == resident and evaluator.deferred_ok(eng)
        _set_roots(eng, envs)
        eng.set_noise_keys()
        eng.simulate(evaluator, sims, use_graph=False)
        visits = eng.root_visits()
        rn, rw = eng.root_stats()
        pri = eng.root_priors()
        eng.check()
        dumps[resident] = (visits.copy(), np.asarray(rw).view(np.uint64).copy(), pri.view(np.uint32).copy())
        eng.close()
        evaluator.hip.close()
    for a, b in zip(dumps[True], dumps[False]):
        assert np.array_equal(a, b)
    assert (dumps[True][0].sum(axis=1) == sims - 1).all() or (dumps[True][0].sum(axis=1) == sims).all()


def test_fp8_selfplay_through_for_network():
    """The product-level opt-in (BatchedSelfPlay.for_network(net_algo='split_f16_fp8')): whole games on two lanes with hipGraphs; the
    games are legal Gomoku games whose pi rows are visit distributions, every lane runs the FP8 kernel's route, and a second run
    reproduces the first bit for bit (the mode is as deterministic as the default)."""
    from rlzero_amd.selfplay import BatchedSelfPlay
    B = 11
    net = _net(B, seed=4, gain=2.0).to('cuda:0').eval()
    runs = []
    for _ in range(2):
        sp = BatchedSelfPlay.for_network(net, B, 5, n_games=16, n_playout=40, device='cuda:0', lanes=2, seed=3, net_algo='split_f16_fp8')
        assert all(lane.evaluator.hip.algo == 'split_f16_fp8' and lane.evaluator.deferred_ok(lane.eng) for lane in sp.lanes)
        trajs = sp.run(range(16))
        for lane in sp.lanes:
            lane.eng.close()
            lane.evaluator.hip.close()
        assert len(trajs) == 16
        for t in trajs:
            e = RefGomoku(B, 5)
            for mv, pi in zip(t.moves, t.pis):
                assert mv in e.leagel_actions() and abs(float(np.sum(pi)) - 1.0) < 1e-5 and pi[mv] > 0
                e.step(int(mv))
            ended, winner = e.game_end_winner()
            assert ended and winner == t.winner
        runs.append({t.game_id: (list(t.moves), [np.asarray(p, dtype=np.float32).tobytes() for p in t.pis]) for t in trajs})
    assert runs[0] == runs[1]
