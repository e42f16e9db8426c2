"""N > 1 path on CPU: two gloo ranks shard the games by id, play them (the oracle stands in
for the device search -- this test is about the sharding and the single gather, not the
kernels), gather to rank 0, and the result equals the single-process run game for game."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO


def _play(game_ids, seed, sims):
    """Trajectories for ``game_ids`` with the counter-based uniforms of rlzero_amd.selfplay."""
    from oracle import evaluators as ev
    from oracle.gomoku_ref import RefGomoku
    from oracle.mcts_ref import RefPlayer, inverse_cdf_choice, self_play_game
    from rlzero_amd.selfplay import Trajectory, move_uniform
    out = []
    for gid in game_ids:
        us = move_uniform(seed, np.full(16, gid), np.arange(16))
        player = RefPlayer(ev.vlin, sims, 5, is_selfplay=True, choice=inverse_cdf_choice(us))
        winner, data, moves = self_play_game(RefGomoku(3, 3), player, temperature=1.0)
        out.append(Trajectory(gid, 3, 3, moves, [pi for _, pi, _ in data], winner))
    return out


def _worker(rank, world, port, n_games, result_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch.distributed as dist
    from rlzero_amd.selfplay import gather_trajectories, shard_game_ids
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = shard_game_ids(n_games, rank, world)
    assert mine == list(range(rank, n_games, world))
    local = _play(mine, seed=3, sims=20)
    merged = gather_trajectories(local, 3, 3, dst=0)
    if rank == 0:
        np.savez(result_path, ids=[t.game_id for t in merged], winners=[t.winner for t in merged],
                 moves=np.concatenate([t.moves for t in merged]),
                 pis=np.concatenate([t.pis for t in merged]))
    else:
        assert merged is None
    # weights broadcast (after policy_update on rank 0): every rank ends with rank 0's parameters
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import broadcast_weights
    torch.manual_seed(100 + rank)
    net = PolicyValueNet(3)
    versions = [p._version for p in net.parameters()]
    broadcast_weights(net, src=0)
    # the copy must be visible to HipNetEvaluator.refresh_if_changed (data_ptr, _version): a write through p.data is not
    assert all(p._version > v for p, v in zip(net.parameters(), versions))
    torch.manual_seed(100)
    want = PolicyValueNet(3)
    for a, b in zip(net.parameters(), want.parameters()):
        assert torch.equal(a, b)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_selfplay_gather_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    n_games = 7  # ragged: rank 0 gets 4 games, rank 1 gets 3
    result = str(tmp_path / 'merged.npz')
    mp.spawn(_worker, args=(2, port, n_games, result), nprocs=2, join=True)
    got = np.load(result)
    single = _play(range(n_games), seed=3, sims=20)
    assert got['ids'].tolist() == list(range(n_games))
    assert got['winners'].tolist() == [t.winner for t in single]
    assert got['moves'].tolist() == [m for t in single for m in t.moves]
    assert np.array_equal(got['pis'], np.concatenate([t.pis for t in single]))


def _edge_worker(rank, world, port, result_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch.distributed as dist
    from rlzero_amd import selfplay
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    calls = []
    real_all_gather, real_gather, real_all_reduce = dist.all_gather, dist.gather, dist.all_reduce
    dist.all_gather = lambda *a, **k: calls.append('all_gather') or real_all_gather(*a, **k)
    dist.gather = lambda *a, **k: calls.append('gather') or real_gather(*a, **k)
    dist.all_reduce = lambda *a, **k: calls.append('all_reduce') or real_all_reduce(*a, **k)
    # (1) a rank with nothing to send, pi as float32: exactly two collectives, one of them the payload
    local = _play([0, 1, 2], seed=3, sims=20) if rank == 0 else []
    merged = selfplay.gather_trajectories(local, 3, 3, dst=0, pi_dtype=np.float32)
    assert calls == ['all_gather', 'gather'] and len(calls) == selfplay.COLLECTIVES_PER_EXCHANGE
    if rank == 0:
        assert [t.game_id for t in merged] == [0, 1, 2]
        for got, want in zip(merged, local):
            assert got.moves == want.moves and got.winner == want.winner
            assert np.array_equal(got.pis, want.pis.astype(np.float32).astype(np.float64))
    # (2) nobody has anything
    assert (selfplay.gather_trajectories([], 3, 3, dst=0) or []) == []
    # (3) packing fails on rank 1 only: BOTH ranks raise, inside the first collective's agreement, nobody waits in the gather
    del calls[:]
    try:
        selfplay.gather_trajectories(_play([4 + rank], seed=3, sims=20) if rank == 0 else [object()], 3, 3, dst=0)
        raised = False
    except RuntimeError as exc:
        raised = 'rank(s) [1]' in str(exc)
    assert raised and calls == ['all_gather']
    dist.barrier()
    open(result_path + '.%d' % rank, 'w').write('ok')
    dist.destroy_process_group()


def test_gather_is_one_size_row_and_one_payload(tmp_path):
    """gather_trajectories = one all_gather of a 3-word size row + ONE gather of a byte payload (header + moves + pi); a rank
    without games takes part with an empty payload; a rank that fails to pack makes every rank raise after the size row."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'edge')
    mp.spawn(_edge_worker, args=(2, port, result), nprocs=2, join=True)
    assert os.path.exists(result + '.0') and os.path.exists(result + '.1')


def test_gather_without_process_group_is_identity():
    from rlzero_amd.selfplay import gather_trajectories, pack_trajectories, unpack_trajectories
    trajs = _play([2, 0, 1], seed=1, sims=10)
    merged = gather_trajectories(trajs, 3, 3)
    assert [t.game_id for t in merged] == [0, 1, 2]
    back = unpack_trajectories(*pack_trajectories(merged, 9), 3, 3)
    for a, b in zip(merged, back):
        assert (a.game_id, a.moves, a.winner) == (b.game_id, b.moves, b.winner)
        assert np.array_equal(a.pis, b.pis)
    w, data = merged[0].as_reference_tuple()
    assert w == merged[0].winner and len(data) == len(merged[0].moves)
    assert data[0][0].shape == (4, 3, 3)


def test_move_uniform_is_a_pure_function_of_seed_game_ply():
    from rlzero_amd.selfplay import move_uniform
    a = move_uniform(7, np.arange(1000), np.zeros(1000, dtype=np.int64))
    b = move_uniform(7, np.arange(1000)[::-1], np.zeros(1000, dtype=np.int64))[::-1]
    assert np.array_equal(a, b) and (a >= 0).all() and (a < 1).all()
    assert abs(a.mean() - 0.5) < 0.05 and len(np.unique(a)) == 1000
    assert move_uniform(7, 3, 4) != move_uniform(8, 3, 4) != move_uniform(7, 4, 3)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with NO external launcher: bench.py starts its two ranks itself (children of a
    process that never touches the GPU), the N > 1 path (barriers around the timed region, MAX over ranks of the
    time, SUM of the work, the trajectory gather, one JSON line from rank 0) runs on a 1-GPU box -- the two ranks
    share cuda:0 and rendezvous over gloo (RCCL refuses two ranks on one device) -- and every game's trajectory
    equals the one the same game id gets in a single-rank run (world-size invariance)."""
    import json
    import subprocess
    env = dict(os.environ, RZ_BENCH_SINGLE_DEVICE='1', RZ_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    dump2, dump1 = str(tmp_path / 'two.json'), str(tmp_path / 'one.json')
    common = ['--steps', '2', '--warmup', '1', '--board', '9', '--playouts', '40', '--no-cpu-baseline',
              '--no-literal-config', '--no-configs']
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--games', '32', '--dump-trajectories',
           dump2] + common
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['scaling'] == 'weak'
    assert rec['config']['games_total'] == 64
    # 2 ranks x 32 games x 2 moves x 40 simulations in the timed region
    assert abs(rec['value'] * rec['ms_per_step'] * 2 / 1000.0 - 2 * 32 * 2 * 40) < 2e-3 * 2 * 32 * 2 * 40   # (ms_per_step is rounded to a microsecond)
    assert rec['cpu_baseline'] is None and rec['fill_1536'] is None
    assert rec['selfplay']['games_sampled'] == 64 and rec['selfplay_games_per_sec'] > 0
    tg = rec['trajectory_gather']  # the one exchange of the path, here over gloo
    assert tg['ranks'] == 2 and tg['games'] >= 64 and tg['unique_game_ids'] and tg['plies'] >= 64 * 9
    assert tg['backend'] == 'gloo' and tg['collectives_per_exchange'] == 2   # one size all_gather + ONE payload gather
    # every rank's own rate is on the line; value = SUM work / MAX seconds lies between world x min and world x max
    rates = rec['per_rank_sims_per_sec']
    assert len(rates) == 2 and min(rates) == rec['per_rank_min'] > 0 and max(rates) == rec['per_rank_max']
    assert 2 * min(rates) * 0.999 <= rec['value'] <= 2 * max(rates) * 1.001
    assert rec['reuse_dropped'] == 0 and rec['selfplay']['reuse_dropped'] == 0
    assert 0 < rec['selfplay']['arena_slots_used_max'] < rec['selfplay']['arena_slots'] == rec['arena_slots']
    assert rec['config']['hw_queues'] == 8
    # the same 64 game ids on ONE rank (one lane of 64 games instead of two ranks x two lanes of 16)
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--games', '64', '--lanes', '1',
           '--dump-trajectories', dump1] + common
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    two, one = json.load(open(dump2)), json.load(open(dump1))
    first_generation = [str(g) for g in range(64)]
    assert all(g in two and g in one for g in first_generation)
    for g in sorted(set(two) & set(one), key=int):
        assert two[g] == one[g], 'game %s depends on the number of ranks' % g
    # a rank that fails makes the self-launched run fail (non-zero exit, no line)
    bad = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--games', '8', '--board', '99']
                         + common[:4], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.decode().splitlines() if ln.startswith('{')]


@pytest.mark.gpu
def test_bench_collectives_on_rccl_with_one_rank():
    """The RCCL side of the N > 1 bench path on a 1-GPU box: one rank, process group 'nccl', every collective of the
    path forced to run (barrier, MAX / SUM all_reduce of float64, all_gather of sizes, gather of trajectories)."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RZ_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '2',
           '--warmup', '1', '--board', '9', '--playouts', '40', '--games', '32', '--no-cpu-baseline',
           '--no-literal-config', '--no-configs']
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')][-1])
    tg = rec['trajectory_gather']
    assert tg.get('error') is None and tg['backend'] == 'nccl' and tg['ranks'] == 1 and tg['collectives_per_exchange'] == 2
    assert len(rec['per_rank_sims_per_sec']) == 1 and abs(rec['per_rank_sims_per_sec'][0] - rec['value']) <= 1e-3 * rec['value']
    assert tg['games'] >= 32 and tg['unique_game_ids'] and tg['payload_bytes'] > 0
    assert rec['selfplay']['games_sampled'] == 32


# ------------------------------------------------------------------ the multi-rank trainer (tools/train_alphazero.py)
def _load_trainer():
    import importlib.util
    spec = importlib.util.spec_from_file_location('train_alphazero', os.path.join(REPO, 'tools', 'train_alphazero.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _train_on_cpu(world, games_in_flight, n_batches):
    """TrainPipeline.run() on the CPU with the oracle as the search (the sequential reference rule driven by the CURRENT
    network: a round's games depend on the weights the rank holds) -> (losses, buffer arrays, final weight sums)."""
    import random
    import torch
    from oracle.evaluators import NetEvaluator
    from oracle.gomoku_ref import RefGomoku
    from oracle.mcts_ref import RefPlayer, inverse_cdf_choice, self_play_game
    from rlzero_amd.selfplay import Trajectory, move_uniform
    torch.set_num_threads(1)
    torch.cuda.is_available = lambda: False   # the trainer picks its device (and the gloo backend) from this
    mod = _load_trainer()
    rank = int(os.environ.get('RANK', '0'))
    random.seed(5)
    np.random.seed(5)
    torch.manual_seed(100 + rank)             # ranks start from DIFFERENT weights: the pipeline must hand out rank 0's
    pipe = mod.TrainPipeline(board_size=3, n_in_row=3, n_playout=20, game_batch_num=n_batches, check_freq=50,
                             selfplay_games_in_flight=games_in_flight, seed=11)
    assert (pipe.rank, pipe.world) == (rank, world) and pipe.selfplay_seed == 11
    pipe.batch_size = 16
    played = []

    def play(game_ids, on_finished=None):   # (TrainPipeline._play_games' signature; the CPU stand-in hands every game over at the end)
        weights = {k: v.detach().cpu().numpy() for k, v in pipe.alphazero_agent.policy_value_net.state_dict().items()}
        evaluator = NetEvaluator(weights, 3)
        out = []
        for gid in game_ids:
            us = move_uniform(pipe.selfplay_seed, np.full(16, gid), np.arange(16))
            player = RefPlayer(evaluator, pipe.n_playout, pipe.c_puct, is_selfplay=True, choice=inverse_cdf_choice(us))
            winner, data, moves = self_play_game(RefGomoku(3, 3), player, temperature=1.0)
            out.append(Trajectory(gid, 3, 3, moves, [pi for _, pi, _ in data], winner))
        played.extend(game_ids)
        if on_finished is not None and out:
            on_finished(out)
        return out

    pipe._play_games = play
    losses = []
    real_update = pipe.policy_update
    pipe.policy_update = lambda: losses.append(real_update()) or losses[-1]
    pipe.run()
    assert played == [g for g in range(n_batches * games_in_flight * world) if g % world == rank]
    buf = list(pipe.data_buffer)
    sums = [float(p.detach().double().abs().sum()) for p in pipe.alphazero_agent.policy_value_net.parameters()]
    return (np.array(losses, dtype=np.float64).reshape(-1, 2),
            np.array([s for s, _, _ in buf], dtype=np.float32).reshape(len(buf), 36),
            np.array([p for _, p, _ in buf], dtype=np.float32).reshape(len(buf), 9),
            np.array([z for _, _, z in buf], dtype=np.float64), np.array(sums))


def _trainer_worker(rank, world, port, result_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), RZ_DIST_BACKEND='gloo')
    import torch.distributed as dist
    losses, states, pis, zs, sums = _train_on_cpu(world, 3, 3)
    if rank == 0:
        np.savez(result_path, losses=losses, states=states, pis=pis, zs=zs, sums=sums)
    else:
        assert len(losses) == 0 and len(zs) == 0   # the buffer and the learner live on rank 0
        np.savez(result_path + '.rank1.npz', sums=sums)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_equals_the_single_process_run(tmp_path):
    """tools/train_alphazero.py on two gloo ranks: every round's games are dealt by id (g mod 2), ONE gather brings the
    trajectories to rank 0 (pi as float32), rank 0 learns, ONE broadcast hands out the weights -- and the losses, the
    replay buffer and the final weights are those of one process playing the same ids (the reference's loop,
    train_alphazero.py:81-137,164-190).  The games depend on the weights, so a rank that missed a broadcast would
    play different games."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'two.npz')
    mp.spawn(_trainer_worker, args=(2, port, result), nprocs=2, join=True)
    two = np.load(result)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    losses, states, pis, zs, sums = _train_on_cpu(1, 6, 3)
    assert len(losses) == 3 and np.array_equal(two['losses'], losses)
    assert np.array_equal(two['states'], states) and np.array_equal(two['pis'], pis) and np.array_equal(two['zs'], zs)
    assert np.array_equal(two['sums'], sums)
    assert np.array_equal(np.load(result + '.rank1.npz')['sums'], sums)   # rank 1 ends with rank 0's weights


@pytest.mark.gpu
def test_trainer_two_ranks_on_one_gpu(tmp_path):
    """`python tools/train_alphazero.py --gpus 2` with NO launcher on a 1-GPU box (the two ranks share cuda:0 and meet over
    gloo: RCCL refuses two ranks on one device): BatchedSelfPlay on each rank's share of the ids, the gather, rank 0's
    policy_update, the weight broadcast, refresh_weights on every lane -- against one process playing the same 16 ids:
    the first round's games are the same (a game depends on (seed, id) only), so its log lines agree.  Three rounds, traced
    (RZ_TRAIN_TRACE): in EVERY round rank 1 holds rank 0's parameters and its HIP evaluators -- the copies of the weights the
    search really runs on -- answer a fixed batch of positions with rank 0's bits, which change from round to round as rank 0
    learns (a rank that kept searching with the weights of round 1 fails here)."""
    import json
    import subprocess
    env = dict(os.environ, RZ_DIST_SINGLE_DEVICE='1', RZ_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    common = ['--board', '6', '--n-in-row', '4', '--playouts', '30', '--batches', '3', '--seed', '3']
    outs, traces = [], []
    for gpus, in_flight in ((2, 8), (1, 16)):
        trace = tmp_path / ('trace%d' % gpus)
        trace.mkdir()
        cmd = [sys.executable, os.path.join(REPO, 'tools', 'train_alphazero.py'), '--gpus', str(gpus), '--games-in-flight',
               str(in_flight)] + common
        out = subprocess.run(cmd, env=dict(env, RZ_TRAIN_TRACE=str(trace)), cwd=str(tmp_path), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=900)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        outs.append([ln for ln in out.stdout.decode().splitlines() if ln.startswith(('batch i:', 'kl:'))])
        traces.append([[json.loads(ln) for ln in open(str(trace / ('rank%d.jsonl' % r)))] for r in range(gpus)])
    two, one = outs
    assert len(two) == 6 and len(one) == 6 and two[0] == one[0] and two[0].startswith('batch i:1, episode_len:')
    num = lambda line: {k: float(v) for k, v in (item.split(':') for item in line.split(','))}  # noqa: E731
    a, b = num(two[1]), num(one[1])
    assert abs(a['loss'] - b['loss']) <= 1e-3 and abs(a['entropy'] - b['entropy']) <= 1e-3
    assert two[2].startswith('batch i:2, episode_len:') and one[2].startswith('batch i:2, episode_len:')
    (rank0, rank1), (single, ) = traces
    assert len(rank0) == len(rank1) == len(single) == 3
    for r0, r1 in zip(rank0, rank1):
        assert r0['params'] == r1['params'], 'round %d: rank 1 does not hold rank 0\'s parameters' % r0['round']
        assert len(set(r0['lanes'] + r1['lanes'])) == 1, 'round %d: an evaluator searches with other weights' % r0['round']
        assert set(r0['games']).isdisjoint(r1['games']) and len(r0['games']) == len(r1['games']) == 8
    assert len({r['lanes'][0] for r in rank1}) == 3 and len({r['params'] for r in rank1}) == 3   # rank 0 learned in between
    # the first round does not depend on the learner: the same games as the single process, id for id
    both = dict(rank0[0]['games'], **rank1[0]['games'])
    assert both == single[0]['games'] and rank0[0]['lanes'][0] == single[0]['lanes'][0]


# ------------------------------------------------------------------ 8-GPU readiness without an 8-GPU box
def _fake_trajectory(gid, board=15, max_plies=None):
    """A finished game made from its id alone (no search): what the sharding and the gather carry, whatever produced it."""
    from rlzero_amd.selfplay import Trajectory
    rs = np.random.RandomState(gid % (2 ** 31))
    cells = board * board
    plies = int(rs.randint(9, 40)) if max_plies is None else int(max_plies)
    moves = rs.permutation(cells)[:plies]
    pis = rs.random_sample((plies, cells))
    pis /= pis.sum(axis=1, keepdims=True)
    return Trajectory(gid, board, 5, moves, pis, int(rs.randint(-1, 2)))


def _world8_worker(rank, world, port, n_games, result_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch
    import torch.distributed as dist
    from rlzero_amd.selfplay import gather_trajectories, shard_game_ids
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = shard_game_ids(n_games, rank, world)
    assert len(mine) in (n_games // world, n_games // world + 1) and all(g % world == rank for g in mine)
    merged = gather_trajectories([_fake_trajectory(g) for g in mine], 15, 5, dst=0, pi_dtype=np.float32)
    if rank == 0:
        np.savez(result_path, ids=[t.game_id for t in merged], winners=[t.winner for t in merged],
                 plies=[len(t.moves) for t in merged], moves=np.concatenate([t.moves for t in merged]),
                 pi_sum=np.array([float(t.pis.astype(np.float64).sum()) for t in merged]), pi0=np.stack([t.pis[0] for t in merged[::97]]))
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_games', [4096, 4099])
def test_world_8_sharding_and_gather_equal_one_process(tmp_path, n_games):
    """BASELINE configs[3] on the CPU: 4096 game ids over EIGHT gloo ranks (g mod 8: 512 each; 4099: three ranks carry 513), every
    rank's finished games to rank 0 in the two collectives of the exchange (pi as float32) == the list one process holds."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'w8.npz')
    mp.spawn(_world8_worker, args=(8, port, n_games, result), nprocs=8, join=True)
    got = np.load(result)
    single = [_fake_trajectory(g) for g in range(n_games)]
    assert got['ids'].tolist() == list(range(n_games))
    assert got['winners'].tolist() == [t.winner for t in single] and got['plies'].tolist() == [len(t.moves) for t in single]
    assert got['moves'].tolist() == [m for t in single for m in t.moves]
    assert np.array_equal(got['pi0'], np.stack([t.pis[0].astype(np.float32).astype(np.float64) for t in single[::97]]))
    assert np.array_equal(got['pi_sum'], np.array([float(t.pis.astype(np.float32).astype(np.float64).sum()) for t in single]))


def _worst_case_worker(rank, world, port, result_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import time
    import torch
    import torch.distributed as dist
    from rlzero_amd.selfplay import gather_trajectories
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    trajs = [_fake_trajectory(rank + world * i, max_plies=225) for i in range(512)]   # 512 games x 225 plies: a rank's worst case
    dist.barrier()
    t0 = time.perf_counter()
    merged = gather_trajectories(trajs, 15, 5, dst=0, pi_dtype=np.float32)
    dist.barrier()
    ms = 1e3 * (time.perf_counter() - t0)
    if rank == 0:
        assert len(merged) == 1024 and [t.game_id for t in merged] == list(range(1024)) and all(len(t.moves) == 225 for t in merged)
        for t in merged[::101]:
            want = _fake_trajectory(t.game_id, max_plies=225)
            assert t.moves == want.moves and np.array_equal(t.pis, want.pis.astype(np.float32).astype(np.float64))
        payload = 512 * 32 + 512 * 225 * (8 + 4 * 225)
        open(result_path, 'w').write('%d %.1f' % (payload, ms))
    dist.destroy_process_group()


def test_gather_at_the_worst_case_payload(tmp_path):
    """ONE exchange at the largest payload a rank of configs[3] can hold -- 512 games x 225 plies, pi as float32: 104.6 MB per
    rank (SURVEY.md 8e sizes it at ~112 MB) -- through the size row + the single payload gather; time printed (gloo over
    loopback here; RCCL over xGMI moves it in about a millisecond per link)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'worst.txt')
    mp.spawn(_worst_case_worker, args=(2, port, result), nprocs=2, join=True)
    payload, ms = open(result).read().split()
    assert int(payload) == 512 * 32 + 512 * 225 * (8 + 4 * 225) == 104_617_984
    print('worst-case gather: %s bytes per rank, two gloo ranks, %s ms' % (payload, ms))


def test_rank_affinity_from_sysfs(tmp_path):
    """rlzero_amd.affinity: a rank's cores = the cores local to ITS GPU's NUMA node, taken from sysfs alone (KFD topology order =
    HIP device order), split evenly among the ranks that share the node; nothing is pinned when sysfs says nothing."""
    from rlzero_amd import affinity
    root = tmp_path / 'sys'
    nodes = root / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes'
    # two CPU nodes, then eight GPUs: the first four on NUMA node 0 (cores 0-63), the others on node 1 (64-127)
    for n in range(10):
        d = nodes / str(n)
        d.mkdir(parents=True)
        gpu = n >= 2
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n' % (0 if gpu else 64, 1024 if gpu else 0,
                                                                                                   128 + n - 2 if gpu else 0))
        if gpu:
            dev = root / 'class' / 'drm' / ('renderD%d' % (128 + n - 2)) / 'device'
            dev.mkdir(parents=True)
            node = 0 if n - 2 < 4 else 1
            (dev / 'numa_node').write_text('%d\n' % node)
            (dev / 'local_cpulist').write_text('0-63\n' if node == 0 else '64-127\n')
    sysfs = str(root)
    assert affinity.parse_cpulist('0-3,8,10-11') == [0, 1, 2, 3, 8, 10, 11] and affinity.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == '0-3,8,10-11'
    gpus = affinity.gpu_numa_nodes(sysfs)
    assert [g[0] for g in gpus] == [0, 0, 0, 0, 1, 1, 1, 1] and gpus[5][1] == list(range(64, 128))
    allowed = list(range(128))
    plans = [affinity.plan(r, list(range(8)), allowed, sysfs, env={}) for r in range(8)]
    assert [p['numa_node'] for p in plans] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert [affinity.format_cpulist(p['cpus']) for p in plans] == ['0-15', '16-31', '32-47', '48-63', '64-79', '80-95', '96-111', '112-127']
    # one rank alone keeps its whole node; a restricted affinity mask is respected; visible-device lists re-map the ordinals
    assert affinity.plan(5, [5], allowed, sysfs, env={})['cpus'] == list(range(64, 128))
    assert affinity.plan(0, [0, 1], list(range(0, 64, 2)), sysfs, env={})['cpus'] == list(range(0, 32, 2))
    assert affinity.plan(0, [0], allowed, sysfs, env={'HIP_VISIBLE_DEVICES': '6,7'})['numa_node'] == 1
    # CUDA_VISIBLE_DEVICES is HIP_VISIBLE_DEVICES' alias, read only when that one is unset -- never a second filter; ROCR_VISIBLE_DEVICES
    # (the runtime below) filters first
    assert affinity.plan(0, [0], allowed, sysfs, env={'HIP_VISIBLE_DEVICES': '6,7', 'CUDA_VISIBLE_DEVICES': '1'})['numa_node'] == 1
    assert affinity.plan(0, [0], allowed, sysfs, env={'CUDA_VISIBLE_DEVICES': '5'})['numa_node'] == 1
    assert affinity.plan(1, [1], allowed, sysfs, env={'ROCR_VISIBLE_DEVICES': '2,3,4,5', 'HIP_VISIBLE_DEVICES': '0,3'})['numa_node'] == 1
    assert affinity.plan(9, [9], allowed, sysfs, env={}) is None
    rec = affinity.pin_to_gpu(2, local_world=8, sysfs=sysfs, env={}, apply=False)
    here = sorted(os.sched_getaffinity(0))   # (this host has fewer cores than the fake node: its share of what both have)
    local = [c for c in range(64) if c in here]
    want = local[2 * (len(local) // 4):3 * (len(local) // 4)] if len(local) >= 4 else local
    assert rec['pinned'] is False and (rec['cpus'] == affinity.format_cpulist(want) if local else 'why' in rec)
    assert sorted(os.sched_getaffinity(0)) == here   # apply=False changes nothing
    assert affinity.pin_to_gpu(0, sysfs=str(tmp_path / 'nothing'), apply=False)['pinned'] is False


@pytest.mark.gpu
def test_bench_four_ranks_on_one_gpu():
    """The N-rank path with more than two ranks, on a 1-GPU box: `bench.py --gpus 4` with its ranks sharing cuda:0 over gloo, 64 games
    per rank (FOUR, not eight: a GPU box of this pool admits at most six processes on its card at once, and the test runner itself
    is one of them -- six ranks were tried and the box's process guard ended the run; the eight-rank run is the driver's, on an
    eight-GPU node, and the world-8 sharding + gather is covered on the CPU above).  Four entries of per-rank rates, every gathered
    game id unique, the two collectives of the exchange, every rank's host affinity on the line."""
    import json
    import subprocess
    env = dict(os.environ, RZ_BENCH_SINGLE_DEVICE='1', RZ_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4', '--games', '64', '--steps', '2', '--warmup', '1', '--board', '9',
           '--playouts', '40', '--regions', '1', '--no-cpu-baseline', '--no-literal-config', '--no-configs']
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert rec['n_gpus'] == 4 and rec['config']['games_total'] == 256
    assert len(rec['per_rank_sims_per_sec']) == 4 and min(rec['per_rank_sims_per_sec']) > 0
    tg = rec['trajectory_gather']
    assert tg['ranks'] == 4 and tg['unique_game_ids'] and tg['collectives_per_exchange'] == 2 and tg['games'] >= 4 * 64
    assert len(rec['host_affinity']) == 4 and all('pinned' in a for a in rec['host_affinity'])
