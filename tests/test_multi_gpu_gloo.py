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
    broadcast_weights(net, src=0)
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
    assert abs(rec['value'] * rec['ms_per_step'] * 2 / 1000.0 - 2 * 32 * 2 * 40) < 1.0
    assert rec['cpu_baseline'] is None and rec['fill_1536'] is None
    assert rec['selfplay']['games_sampled'] == 64 and rec['selfplay_games_per_sec'] > 0
    tg = rec['trajectory_gather']  # the one exchange of the path, here over gloo
    assert tg['ranks'] == 2 and tg['games'] >= 64 and tg['unique_game_ids'] and tg['plies'] >= 64 * 9
    assert tg['backend'] == 'gloo'
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
    assert tg.get('error') is None and tg['backend'] == 'nccl' and tg['ranks'] == 1
    assert tg['games'] >= 32 and tg['unique_game_ids'] and tg['payload_bytes'] > 0
    assert rec['selfplay']['games_sampled'] == 32
