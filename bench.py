#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations / second (and self-play games / second) of AlphaZero self-play on
15x15 Gomoku, 800 simulations per move (BASELINE.json configs[3]), random-init PolicyValueNet
(torch.manual_seed(0)), fp32.  Games in flight per GPU are an engine parameter (the batch of the leaf
evaluation): the default keeps 2 lanes x 768 games = 1536 per GPU, which is what fills an MI355X -- the
network trunk of one lane runs as 256 persistent workgroups (one per CU, three boards each) while the tree / FC
kernels of the other lane run beside it on the same CUs (their waves fit next to a resident trunk workgroup).
`--games 512` is the literal 4096 / 8 games per GPU of configs[3]; at N = 1 the default run measures it too
(`literal_config`); `--trunk-wgs 224 --games 1344` is the capped layout of round 1.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one move of every game on the GPU: n_playout simulation steps (select ->
evaluate -> expand/backup for all games), pi from the root visits, a move drawn and applied,
tree reuse; finished games are replaced by fresh ones so the batch stays full.  Games are
independent, so N GPUs play N x 1536 games with no collective in the timed region (weak
scaling); rank 0 prints ONE JSON line.

Also on the line:
  roofline      for the dominant kernel region (the policy+value forward of the leaf batch,
                the path's one dense contraction): algorithmic FLOPs F(S) = 188416*S + 8*S^2 +
                128 per position (the trunk kernel alone: 188160*S) x positions per launch /
                its average duration from HIP events on the launch streams (two lanes: the union of their
                intervals / launches), against the peak of the pipe it runs on: the f16 MFMA peak / 3 for the
                default trunk (three f16 MFMAs per f32 product), the f32-input MFMA peak for the others.
  selfplay      after the timed steps the first-generation games are played to their end (slots refilled):
                games/s = moves/s of that leg / mean plies per game.
  cpu_baseline  the oracle (Python restatement of the reference, batch-1 torch CPU forward,
                one thread per process) timed on this host's cores on a bounded sample of
                the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BOARD, N_ROW, N_PLAYOUT, GAMES_PER_GPU, C_PUCT, TEMPERATURE = 15, 5, 800, 512, 5.0, 1.0
RESERVED_CUS_PER_XCD, N_XCD = 4, 8  # --trunk-wgs 224: CUs a capped trunk leaves to the other lane's small kernels
BOARDS_PER_WORKGROUP = 3  # boards per persistent trunk workgroup and step at the default batch
PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MATRIX_TFLOPS = 2500.0  # dense f16 / bf16 MFMA, same table
SPLIT_MFMAS_PER_PRODUCT = 3      # split_f16: hi*hi + hi*lo + lo*hi
PEAK_HBM_GBS = 8000.0
GATHER_SAMPLE_GAMES = 256  # N > 1: finished games per rank sent to rank 0 by the trajectory gather (outside the timed region)


def flops_per_position(cells):
    return 188416 * cells + 8 * cells * cells + 128


def trunk_flops_per_position(cells):
    """conv1 + conv2 + conv3 + the two 1x1 head convolutions: 2*(9*(4*32 + 32*64 + 64*128) + 128*6) per cell."""
    return 188160 * cells


def executed_flop_ratio(args, cells):
    """Flops the matrix pipe executes / algorithmic trunk flops (15x15).  f32-MFMA kernels: MFMAs of 2048 flops
    per board -- direct 21870 (rows x 16 columns tiles), Winograd F(4x4,3x3) 6030, the 1x1 heads on the VALU.  split_f16: 4320 f16 MFMAs of 32768 flops (3 per product, 32-position tiles on 2 x 15
    columns) + conv1's 270 f32 MFMAs."""
    if args.evaluator != 'hipnet' or args.game != 'gomoku' or args.board != 15:
        return 1.0
    mfmas = {'winograd_f4': 6030, 'direct': 21870, 'split_f16': 4050 * 16 + 270, 'split_f16_tiles': 4320 * 16 + 270}[args.net_algo]
    return mfmas * 2048.0 / trunk_flops_per_position(cells)


def trunk_peak(args):
    """-> (peak TFLOP/s the trunk's ALGORITHMIC flops are priced against, peak of the pipe it executes on, note)."""
    if args.evaluator == 'hipnet' and args.net_algo.startswith('split_f16'):
        return (PEAK_F16_MATRIX_TFLOPS / SPLIT_MFMAS_PER_PRODUCT, PEAK_F16_MATRIX_TFLOPS,
                'peak = dense f16 MFMA peak (2500 TFLOP/s) / 3: every f32 product costs three f16 MFMAs (operands '
                'carried as hi + lo f16 pairs, f32 accumulation; error at the level of the exact-f32 kernel, '
                'tests/test_gpu_parity.py); on random operands the chip sustains ~1590 TFLOP/s of f16 MFMA under its '
                'power limit (profiles/r01/f16_mfma_rate.txt), i.e. ~530 TFLOP/s of such products')
    return (PEAK_FP32_MATRIX_TFLOPS, PEAK_FP32_MATRIX_TFLOPS, 'peak = dense f32-input MFMA peak')


def tree_bytes_per_sim(scanned, created, depth):
    """12 B per scanned child, 16 B per created child, 24 B per backed-up node,
    64 B of root bitboards."""
    return 12.0 * scanned + 16.0 * created + 24.0 * (depth + 1.0) + 64.0


def pmc_traffic(kernel, workload, lanes):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes (separate
    FETCH_SIZE / WRITE_SIZE runs, gfx950 read correction applied; profiles/r02/pmc_traffic.json).
    None when no counter run exists for this workload / launch geometry."""
    for rnd in ('r02', 'r01'):  # the newest committed counter run that matches this launch geometry
        try:
            rec = json.load(open(os.path.join(REPO, 'profiles', rnd, 'pmc_traffic.json')))
            if rec['workload'] != workload or rec.get('lanes', 1) != lanes:
                continue
            return rec['kernels'][kernel]['traffic_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            continue
    return None


# --------------------------------------------------------------------------- CPU baseline
def cpu_worker(seconds, game, board, n_playout):
    """One process of the CPU baseline: the oracle plays self-play moves of the same configuration in reference
    mode -- one simulation at a time, batch-1 torch CPU forward, moves drawn the way the reference draws them
    (softmax(log(N + 1e-10) / T) over the root visits, numpy.random.choice; alphazero_mcts.py:88-92,148)."""
    import numpy as np
    import torch
    torch.set_num_threads(1)
    np.random.seed(os.getpid() % 65536)
    if game == 'muzero':
        return cpu_worker_muzero(seconds, n_playout)
    from oracle.evaluators import NetEvaluator
    from oracle.mcts_ref import RefSearch, softmax
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    torch.manual_seed(0)
    if game == 'connect4':
        from oracle.connect4_ref import RefConnect4
        net, env, shape = PolicyValueNet(6, 7, 7), RefConnect4(6, 7, 4), (6, 7)
    else:
        from oracle.gomoku_ref import RefGomoku
        net, env, shape = PolicyValueNet(board), RefGomoku(board, N_ROW if board >= 5 else board), board
    weights = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    search = RefSearch(NetEvaluator(weights, shape), n_playout, C_PUCT)
    sims = 0
    t0 = time.perf_counter()
    deadline = t0 + seconds
    chunk = min(50, n_playout)
    with torch.no_grad():
        while time.perf_counter() < deadline:
            done = 0
            while done < n_playout and time.perf_counter() < deadline:  # in chunks: the budget holds on slow hosts
                m = min(chunk, n_playout - done)
                for _ in range(m):
                    search.playout(env.clone())
                done += m
                sims += m
            if done < n_playout:
                break
            acts = search.root.acts
            visits = np.array([k.n for k in search.root.kids])
            probs = softmax(1.0 / TEMPERATURE * np.log(visits + 1e-10))
            move = int(np.random.choice(acts, p=probs))
            search.update_with_move(move)
            env.step(move)
            if env.game_end_winner()[0]:
                env.reset()
                search.update_with_move(-1)
    print(json.dumps({'sims': sims, 'seconds': time.perf_counter() - t0}))


def cpu_worker_muzero(seconds, n_sims):
    """configs[4] on the CPU: the restated MuZero pseudocode (oracle/muzero_ref.py) on a scalar CartPole-v1 with the
    same random-init MLPs evaluated at batch 1 on torch CPU."""
    import numpy as np
    import torch
    from oracle import muzero_ref as mz
    from rlzero_amd.muzero import MuZeroNet
    torch.manual_seed(0)
    net = MuZeroNet().eval()
    cfg = mz.MuZeroConfig(num_simulations=n_sims)

    def recurrent(hidden, action, path):
        nxt, reward, logits, value = net.recurrent_inference(hidden, torch.tensor([action]))
        return nxt, float(reward), torch.softmax(logits, dim=1)[0].tolist(), float(value)

    env = mz.RefCartPole()
    rs = np.random.RandomState(os.getpid() % 65536)
    state = env.reset(rs.uniform(-0.05, 0.05, 4))
    sims, t0 = 0, time.perf_counter()
    deadline = t0 + seconds
    with torch.no_grad():
        while time.perf_counter() < deadline:
            root = mz.Node(0)
            s0, logits, _ = net.initial_inference(torch.tensor([state], dtype=torch.float32))
            mz.expand_node(root, s0, 0.0, torch.softmax(logits, dim=1)[0].tolist())
            mz.add_exploration_noise(cfg, root, rs.dirichlet([cfg.root_dirichlet_alpha] * 2))
            mz.run_mcts(cfg, root, recurrent)
            sims += n_sims
            visits = np.array([c.visit_count for c in root.children], dtype=np.float64)
            action = int(rs.choice(2, p=visits / visits.sum()))
            state, _, terminated, truncated = env.step(action)
            if terminated or truncated:
                state = env.reset(rs.uniform(-0.05, 0.05, 4))
    print(json.dumps({'sims': sims, 'seconds': time.perf_counter() - t0}))


def usable_cores():
    """Cores this process may really use: min(visible, affinity, cgroup v2 cpu.max quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def run_cpu_baseline(seconds, game='gomoku', board=BOARD, n_playout=N_PLAYOUT):
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', str(seconds), '--game', game, '--board', str(board),
           '--playouts', str(n_playout)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=REPO)
             for _ in range(cores)]
    total, worst = 0, 0.0
    for p in procs:
        out, _ = p.communicate()
        try:
            rec = json.loads(out.decode().strip().splitlines()[-1])
            total += rec['sims']
            worst = max(worst, rec['seconds'])
        except Exception:  # noqa: BLE001
            pass
    if worst <= 0:
        return None
    what = {'gomoku': '%dx%d Gomoku self-play' % (board, board), 'connect4': '6x7 Connect4 self-play',
            'muzero': 'MuZero CartPole-v1 episodes'}[game]
    return {'value': round(total / worst, 1), 'unit': 'sims/s', 'cores': cores, 'kind': 'port',
            'per_process': round(total / worst / cores, 1),
            'sample': '%d processes x %.0f s of %s at %d sims/move from the start position, oracle (%s) + batch-1 '
                      'torch CPU forward, 1 thread each, moves sampled like the reference (numpy.random.choice on '
                      'softmax(log N)); the port runs ~2x the reference\'s own 376-444 sims/s/thread at 15x15 '
                      '(BASELINE.md section 2): it is the faster of the two CPU paths'
                      % (cores, seconds, what, n_playout, 'oracle/muzero_ref.py' if game == 'muzero' else 'oracle/mcts_ref.py')}


def child_line(flags, timeout=900):
    """Run `bench.py <flags>` as a child process (this process has not touched the GPU) -> its JSON line or None."""
    cmd = [sys.executable, os.path.abspath(__file__)] + [str(f) for f in flags]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=REPO, timeout=timeout).stdout
        return json.loads([ln for ln in out.decode().splitlines() if ln.startswith('{')][-1])
    except Exception:  # noqa: BLE001
        return None


def run_literal_config(args):
    """The literal share of configs[3], 512 games in flight, as a child process with the lane layout plan_lanes() picks
    for that batch (two lanes of 256 games, un-capped trunks, 'parts' FC GEMM) -> the fields of its line worth keeping."""
    # (a few more warm-up moves than the main run: the GPU has idled through the CPU baseline before this child starts)
    rec = child_line(['--lanes', 2, '--games', GAMES_PER_GPU, '--trunk-wgs', 0, '--steps', args.steps, '--warmup',
                      max(args.warmup, 4), '--net-algo', args.net_algo, '--graph', args.graph, '--noise', args.noise,
                      '--no-cpu-baseline', '--no-games-leg', '--no-literal-config', '--no-configs'], 600)
    if rec is None:
        return None
    rf = rec.get('roofline') or {}
    return {'workload': rec['config']['workload'], 'lanes': 2, 'value': rec['value'], 'unit': rec['unit'],
            'ms_per_step': rec['ms_per_step'],
            # all trunk flops of the run / wall-clock, and the trunk launched alone (256 boards: one round on 256 CUs); the
            # event-bracketed launches of this layout contain the wait for the other lane's trunk (eager samples)
            'roofline_frac': rf.get('whole_job_frac'), 'roofline_exclusive_frac': rf.get('exclusive_frac'),
            'roofline_exclusive_launch_ms': rf.get('exclusive_launch_ms'),
            'note': 'same engine, %d games in flight = 4096 games / 8 GPUs (BASELINE.json configs[3]) as two lanes of 256 '
                    'with un-capped trunks, the FC GEMM and the tree step of one lane co-resident with the other lane\'s '
                    'trunk (selfplay.plan_lanes); measured by a child process before the main run; one lane of 512 '
                    '(--lanes 1 --games 512) runs 3-12 %% below it (profiles/r02/lane_sweeps.txt)' % GAMES_PER_GPU}


# the other configurations of BASELINE.json, each measured by a child process of the default N = 1 run
CONFIG_LEGS = (  # (key, title, flags, seconds of CPU baseline at --cpu-seconds 60); a few warm-up moves each: a leg starts on a GPU
    # that has idled through its own CPU baseline (the main run adds up to 3 moves to its W until 0.3 s have passed)
    ('C1', 'configs[0] TicTacToe, 25 sims/move, 1 game', ['--board', 3, '--playouts', 25, '--games', 1, '--lanes', 1, '--steps', 9, '--warmup', 20], 5.0),
    ('C1_16_games', 'configs[0] with as many games as the CPU baseline plays at once (16 processes = 16 games): like for like with '
     'its aggregate figure', ['--board', 3, '--playouts', 25, '--games', 16, '--lanes', 1, '--steps', 9, '--warmup', 20, '--no-cpu-baseline'], 0.0),
    ('C2', 'configs[1] 9x9 Gomoku, 200 sims/move, 64 games', ['--board', 9, '--playouts', 200, '--games', 64, '--lanes', 1, '--steps', 8, '--warmup', 8], 12.0),
    ('C2_16_in_flight', 'configs[1] with the opt-in virtual-loss mode: 16 simulations in flight per tree (NOT the reference\'s '
     'sequential search; leaf batches of 1024 instead of 64)',
     ['--board', 9, '--playouts', 200, '--games', 64, '--lanes', 1, '--steps', 8, '--warmup', 8, '--in-flight', 16, '--no-cpu-baseline'], 0.0),
    ('C3', 'configs[2] Connect4, 400 sims/move, 512 games', ['--game', 'connect4', '--playouts', 400, '--games', 512, '--lanes', 2, '--steps', 6, '--warmup', 6], 12.0),
    ('C5', 'configs[4] MuZero CartPole-v1, 50 sims/move, 8192 environments (two 16-environment workgroups per CU)',
     ['--game', 'muzero', '--playouts', 50, '--games', 8192, '--steps', 512, '--warmup', 48], 12.0),
)


def run_config_legs(args):
    out = {}
    for key, title, flags, cpu_s in CONFIG_LEGS:
        rec = child_line(flags + ['--cpu-seconds', max(1.0, cpu_s * args.cpu_seconds / 60.0), '--no-games-leg',
                                  '--no-literal-config', '--no-configs'] + (['--no-cpu-baseline'] if args.no_cpu_baseline else []), 600)
        if rec is None:
            out[key] = {'config': title, 'error': 'the child process printed no line'}
            continue
        rf = rec.get('roofline')
        if rf:
            rf = {k: rf[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms',
                                     'exclusive_frac', 'eager_samples_host_bound', 'note') if k in rf}
        out[key] = {'config': title, 'workload': rec['config']['workload'], 'value': rec['value'], 'unit': rec['unit'],
                    'ms_per_step': rec['ms_per_step'], 'steps': rec['steps'], 'roofline': rf,
                    'cpu_baseline': rec.get('cpu_baseline')}
        for extra in ('multi_sim',):
            if extra in rec.get('config', {}):
                out[key][extra] = rec['config'][extra]
    return out


def launch_ranks(n):
    """Run this very command line under `python -m torch.distributed.run --nproc-per-node n` as a CHILD process
    (this process has not initialised the GPU and never will), pass its stdout / stderr through and return its exit
    code: non-zero if any rank failed."""
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env, cwd=REPO).returncode


# --------------------------------------------------------------------------- GPU run
class TimedEvaluator(object):
    """Brackets every evaluator call (the policy+value forward of the leaf batch) with HIP
    events on the launch stream (torch's current stream is the stream our kernels use)."""

    @property
    def needs_obs(self):
        return getattr(self.inner, 'needs_obs', True)

    def _trunk(self, eng):
        hip = self.inner.hip
        if self.needs_obs:
            hip.trunk_internal(eng.obs)
        else:
            hip.trunk_leaves(eng)  # the leaf bitboards, no float planes

    def __init__(self, inner, torch, label):
        self.inner, self.torch, self.label = inner, torch, label
        self.events = []
        self.fc_events, self.tree_events, self.last_c = [], [], None
        self.record = False

    def __call__(self, eng):
        t = self.torch
        if not self.record:
            return self.inner(eng)
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        hip = getattr(self.inner, 'hip', None)
        if hip is not None:  # bracket the dominant kernel (k_trunk) alone
            a.record()
            hip.trunk_internal(eng.obs)
            b.record()
            out = hip.heads(eng.obs.shape[0], eng.logp, eng.value)  # (the un-fused route reads float planes)
        else:
            a.record()
            out = self.inner(eng)
            b.record()
        self.events.append((a, b))
        return out

    def begin_chunk(self):
        """engine.sim_chunk starts a chunk of eager simulations: the previous chunk's last event pairs with nothing."""
        self.last_c = None

    @property
    def fused_heads(self):
        return getattr(self.inner, 'fused_heads', False)

    def raw_heads(self, eng):
        """Fused route (the tree kernel finishes the heads): k_trunk bracketed when recording; a third event behind
        the FC GEMM brackets that kernel and, with the first event of the NEXT simulation of the same eager chunk,
        the tree step launched in between (same stream)."""
        if not self.record:
            self.last_c = None
            return self.inner.raw_heads(eng)
        t = self.torch
        a, b, c = (t.cuda.Event(enable_timing=True) for _ in range(3))
        hip = self.inner.hip
        a.record()
        if getattr(self, 'last_c', None) is not None:
            self.tree_events.append((self.last_c, a))
        self._trunk(eng)
        b.record()
        self.events.append((a, b))
        out = hip.heads_gemm(eng.obs.shape[0])
        c.record()
        self.fc_events.append((b, c))
        self.last_c = c
        return out

    def mean_ms(self):
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)


MZ_FLOPS_PER_SIM = 2 * (66 * 64 + 3 * 64 * 64 + 64 + 2 * 64 + 64)  # recurrent inference: dyn1, dyn2, rew1, pre1, rew2, pol, val
MZ_BYTES_PER_SIM = 2 * 256 + 40 * 4 + 3 * 4 + 2 * 4 + 8  # hidden state read + written (64 f32), ~4 tree nodes x 40 B, indices, policy, reward / value


def run_muzero(args, rank, world, device, dist, red_device, use_dist=False, cpu_baseline=None):
    """BASELINE.json configs[4]: MuZero on CartPole-v1, 50 simulations per move (random-init model).  A step =
    one move of every environment: initial inference, n simulations (HIP tree kernels + recurrent inference on
    the batch), action sampling, environment step."""
    import torch
    from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
    G = args.games if args.games > 0 else 8192
    n_sims = args.playouts if args.playouts != N_PLAYOUT else 50
    torch.manual_seed(0)
    net = MuZeroNet().to(device).eval()
    sp = MuZeroSelfPlay(net, CartPoleBatch(G, device, seed=rank), n_sims=n_sims, seed=rank, fused=bool(args.mz_fused))
    if args.mz_gpw:
        sp.tree.set_search_shape(args.mz_gpw)
    sp.collect(args.warmup)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    sims0 = sp.sims_done
    sp.sim_events = []  # HIP events around every 8th simulation step (one hipGraph replay) of the timed region
    if sp.fused:
        sp.search_events = []  # ... or around every fused search launch (all simulations of a move)
    t0 = time.perf_counter()
    finished = len(sp.collect(args.steps))  # (fused moves: launches of sp.moves_per_launch moves, records read a launch behind)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    total = float(sp.sims_done - sims0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([total], dtype=torch.float64, device=red_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total = float(c.item())
    sp.tree.check()
    if rank == 0:
        roofline = None
        events = sp.search_events if sp.fused else sp.sim_events
        per_launch = G * (n_sims if sp.fused else 1)  # simulations one launch carries
        if events:
            moves_per_launch = sum(e[2] for e in events) / len(events) if sp.fused_moves else 1
            per_launch = per_launch * moves_per_launch
            ms = sum(e[0].elapsed_time(e[1]) for e in events) / len(events)
            gbs = MZ_BYTES_PER_SIM * per_launch / (ms * 1e-3) / 1e9
            tf = MZ_FLOPS_PER_SIM * per_launch / (ms * 1e-3) / 1e12
            if sp.fused:
                roofline = {'bound': 'mfma', 'kernel': sp.sim_step_label + ', %d environments per launch' % G,
                            'achieved': round(tf, 3), 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': round(tf / PEAK_FP32_MATRIX_TFLOPS, 5), 'traffic': None, 'avg_launch_ms': round(ms, 4),
                            'launches_timed': len(events), 'hbm_achieved_gbs': round(gbs, 2),
                            'moves_per_launch': moves_per_launch,
                            'note': 'achieved = algorithmic flops of the recurrent inference (%d per simulation: its 7 dense '
                                    'layers) x %d simulations x %d environments x %g moves per launch / launch duration (HIP events), '
                                    'against the f32-input MFMA peak (157.3 TFLOP/s); %d workgroups of 16 environments (4 waves: '
                                    'v_mfma_f32_16x16x4_f32 tiles, weights in registers, trees in LDS); a simulation is a chain of '
                                    'dependent steps per tree (fp64 walk, gather, 4 layers, backup): latency bound, not a roof'
                                    % (MZ_FLOPS_PER_SIM, n_sims, G, moves_per_launch, (G + 15) // 16)}
            else:
                roofline = {'bound': 'hbm', 'kernel': sp.sim_step_label + ', %d environments per launch' % G,
                            'achieved': round(gbs, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5),
                            'traffic': None, 'avg_launch_ms': round(ms, 4), 'launches_timed': len(events),
                            'flops_achieved_tflops': round(tf, 3), 'flops_frac_of_f32_mfma_peak': round(tf / PEAK_FP32_MATRIX_TFLOPS, 5),
                            'note': 'one simulation step of all environments (select -> recurrent inference -> expand + backup), '
                                    'HIP events on the launch stream; algorithmic bytes per simulation = %d (hidden state in + out, '
                                    'tree nodes on the path, network heads), algorithmic flops = %d (the 7 dense layers of the '
                                    'recurrent inference): arithmetic intensity 45 flop/B is above the ridge only nominally -- at '
                                    '4096 environments the step is bound by launch / dependent latency, not by either roof'
                                    % (MZ_BYTES_PER_SIM, MZ_FLOPS_PER_SIM)}
        print(json.dumps({
            'metric': 'mcts_sims_per_sec', 'value': round(total / elapsed, 1), 'unit': 'sims/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000.0 * elapsed / max(args.steps, 1), 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 model / f64 tree',
            'data': 'synthetic (random-init MuZero MLPs, torch.manual_seed(0); CartPole-v1 restatement)',
            'config': {'workload': 'muzero_cartpole_v1_%dsims_per_move_%denvs_per_gpu' % (n_sims, G),
                       'games_total': G * world, 'discount': 0.997, 'parallelism': 'environments sharded, dp%d' % world},
            'episodes_finished_in_timed_region': finished, 'tree_hbm_bytes': int(sp.tree.device_bytes),
            'roofline': roofline, 'cpu_baseline': cpu_baseline}), flush=True)
    sp.close()
    if use_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--cpu-worker', type=float, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-seconds', type=float, default=60.0,
                    help='wall-clock budget of the CPU baseline of the main workload (SURVEY.md 8d: >= 60 s); the '
                         'baselines of the `configs` legs get 5-12 s each, scaled with this value')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the child-process legs for the other BASELINE.json configurations (C1, C2, C3, C5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--games', type=int, default=0,
                    help='games per GPU; 0 = lanes x %d boards x trunk workgroups (1536) with 2 lanes, %d with 1' %
                    (BOARDS_PER_WORKGROUP, GAMES_PER_GPU))
    ap.add_argument('--trunk-wgs', type=int, default=0,
                    help='persistent trunk workgroups per lane; 0 = one per CU (default: with lanes > 1 the small kernels of '
                         'one lane run beside the other lane\'s trunk on the same CUs); %d = the capped layout of round 1 '
                         '(4 CUs per XCD left to the small kernels)' % (256 - RESERVED_CUS_PER_XCD * N_XCD))
    ap.add_argument('--board', type=int, default=BOARD)
    ap.add_argument('--game', default='gomoku', choices=['gomoku', 'connect4', 'muzero'],
                    help='connect4: 6x7, 4 in a row, 7 column actions (BASELINE config 3; pair with '
                         '--playouts 400 --games 512); muzero: learned-dynamics MCTS on CartPole-v1 (BASELINE config 5; '
                         'pair with --playouts 50 --games 4096)')
    ap.add_argument('--playouts', type=int, default=N_PLAYOUT)
    ap.add_argument('--evaluator', default='hipnet', choices=['hipnet', 'torchnet', 'vlin'],
                    help="hipnet: hand-written fused fp32 MFMA forward (csrc/rz_net.hip); torchnet: "
                         "PyTorch-ROCm/MIOpen; vlin: synthetic evaluator (isolates the tree kernels)")
    ap.add_argument('--graph', type=int, default=8, help='simulation steps per hipGraph (0 = eager)')
    ap.add_argument('--net-algo', default='split_f16', choices=['winograd_f4', 'direct', 'split_f16', 'split_f16_tiles'])
    ap.add_argument('--no-games-leg', action='store_true',
                    help='skip the self-play games/s leg (after the timed steps the games of the first generation '
                         'are played to their end, slots refilled, to measure moves/s over whole games and the mean '
                         'game length)')
    ap.add_argument('--no-literal-config', action='store_true',
                    help='skip the extra N=1 measurement of the literal configs[3] share: 1 lane x %d games' % GAMES_PER_GPU)
    ap.add_argument('--heads-algo', default='auto', choices=['auto', 'f32', 'split32', 'split64', 'parts'],
                    help='GEMM of the first FC layers (rz_net_set_heads_algo)')
    ap.add_argument('--noise', type=int, default=1,
                    help='Dirichlet(0.3) noise mixed into the priors of EVERY expanded node, as the reference does in '
                         'self-play (node.py:63-69, alphazero_mcts.py:124-129); under its UCT rule the priors are never '
                         'read, so this is work with no effect on the moves -- kept because the reference does it')
    ap.add_argument('--dump-trajectories', default='',
                    help='rank 0 writes the finished games it holds after the run (N > 1: the gathered ones) as JSON '
                         '{game id: moves, winner, first pi}: a game must not depend on the number of ranks')
    ap.add_argument('--eager-every', type=int, default=10,
                    help='two extra moves AFTER the timed region launch every k-th graph chunk kernel by kernel with HIP events '
                         'around the kernels (the timing samples behind roofline.avg_launch_ms of a one-lane run and '
                         'small_kernels); 0 = none')
    ap.add_argument('--pipeline', type=int, default=1,
                    help='1 = the host side of a lane\'s move runs under the other lanes\' simulations (BatchedSelfPlay.'
                         'play_move_pipelined); 0 = all lanes simulate, then all are finished on the host')
    ap.add_argument('--mz-gpw', type=int, default=0,
                    help='--game muzero: games per workgroup of k_mz_search (rz_mz_set_search_shape; 0 = automatic)')
    ap.add_argument('--mz-fused', type=int, default=1,
                    help='--game muzero: 1 = the whole search of a move in one kernel launch (k_mz_search), 0 = one hipGraph '
                         'of tree kernels + PyTorch-ROCm layers per simulation')
    ap.add_argument('--in-flight', type=int, default=1,
                    help='K > 1: opt-in virtual-loss mode, K simulations of every tree share one evaluator batch (NOT the '
                         'reference\'s sequential search: results differ from it; for batches too small to fill the GPU)')
    ap.add_argument('--lanes', type=int, default=2,
                    help='independent batches of games on separate HIP streams (the tree / FC kernels of one '
                         'lane run beside the network trunk of the other)')
    args = ap.parse_args()
    if args.cpu_worker is not None:
        cpu_worker(args.cpu_worker, args.game, args.board, args.playouts)
        return

    # `python bench.py --gpus N` without a launcher: start the N ranks as children (one process per GPU,
    # torch.distributed.run) BEFORE anything touches the GPU, relay rank 0's line, exit with their code
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU
    cpu_baseline = None
    default_config = (args.game == 'gomoku' and args.board == BOARD and args.playouts == N_PLAYOUT)
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        playouts = 50 if (args.game == 'muzero' and args.playouts == N_PLAYOUT) else args.playouts
        cpu_baseline = run_cpu_baseline(args.cpu_seconds, args.game, args.board, playouts)

    # the literal share of configs[3] (4096 games / 8 GPUs = 512 per GPU, one lane) in a child process of its
    # own, also before this process touches the GPU (a process that has initialised the GPU starts no program)
    literal = None
    if (world == 1 and args.gpus == 1 and not args.no_literal_config and args.game == 'gomoku' and args.games == 0
            and args.board == BOARD and args.playouts == N_PLAYOUT and args.evaluator == 'hipnet'):
        literal = run_literal_config(args)
    # ... and the other configurations of BASELINE.json (C1, C2, C3, C5), each a child process with its own
    # roofline and CPU baseline: reported under `configs` of the default N = 1 line
    config_legs = None
    if (world == 1 and args.gpus == 1 and not args.no_configs and default_config and args.games == 0
            and args.evaluator == 'hipnet'):
        config_legs = run_config_legs(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from rlzero_amd.engine import HipNetEvaluator, MCTSEngine, NetEvaluator, SyntheticEvaluator
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.selfplay import BatchedSelfPlay

    # test hooks: run several ranks on ONE GPU with gloo (the multi-rank code path on a 1-GPU box)
    if os.environ.get('RZ_BENCH_SINGLE_DEVICE') == '1':
        local_rank = 0
    backend = os.environ.get('RZ_BENCH_BACKEND', 'nccl')
    # RZ_BENCH_FORCE_DIST=1: initialise the process group and run every collective of the N > 1 path even at world
    # size 1 -- exercises the RCCL calls (barrier, MAX / SUM all_reduce, all_gather + gather of trajectories) on a 1-GPU box
    use_dist = world > 1 or os.environ.get('RZ_BENCH_FORCE_DIST') == '1'
    torch.cuda.set_device(local_rank)
    device = 'cuda:%d' % local_rank
    red_device = device if backend == 'nccl' else 'cpu'  # where the MAX / SUM reductions of the result live
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    if args.game == 'muzero':
        run_muzero(args, rank, world, device, dist, red_device, use_dist, cpu_baseline)
        return
    board, n_row = args.board, (N_ROW if args.board >= 5 else args.board)
    cells = board * board
    if args.game == 'connect4':
        board, n_row, cells = (6, 7), 4, 42
    n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    lanes = max(1, args.lanes)
    trunk_wgs = max(0, args.trunk_wgs)
    # default batch: lanes x 3 boards x trunk workgroups (1536 with two un-capped lanes on 256 CUs)
    G = args.games if args.games > 0 else \
        (lanes * BOARDS_PER_WORKGROUP * (trunk_wgs if trunk_wgs > 0 else n_cus) if lanes > 1 and args.evaluator == 'hipnet'
         else GAMES_PER_GPU)
    heads_algo = args.heads_algo
    if heads_algo == 'auto' and lanes > 1 and trunk_wgs == 0 and args.evaluator == 'hipnet' and args.net_algo.startswith('split_f16'):
        heads_algo = 'parts'  # un-capped lanes: the LDS-free GEMM that fits beside a resident trunk workgroup
    lanes = max(1, min(lanes, G))
    per_lane = [G // lanes + (1 if i < G % lanes else 0) for i in range(lanes)]
    torch.manual_seed(0)  # identical weights on every rank
    net = (PolicyValueNet(6, 7, 7) if args.game == 'connect4' else PolicyValueNet(board)).to(device).eval()
    net_shape = (6, 7, 7) if args.game == 'connect4' else board
    engines, evaluators = [], []
    for g_lane in per_lane:
        eng = MCTSEngine(board, n_row, n_games=g_lane, n_playout=args.playouts, c_puct=C_PUCT, device=device,
                         game=args.game, add_noise=bool(args.noise), noise_seed=1000 * rank + len(engines),
                         sims_in_flight=args.in_flight)
        if args.evaluator == 'hipnet':
            hip_ev = HipNetEvaluator(net, net_shape, device, max_boards=eng.n_leaves)
            hip_ev.hip.set_algo(args.net_algo)
            hip_ev.hip.set_heads_algo(heads_algo)
            hip_ev.hip.set_max_workgroups(trunk_wgs)
            ev = TimedEvaluator(hip_ev, torch,
                                {'winograd_f4': 'k_trunk_wino_f4<4> (hand-written fused fp32-MFMA conv trunk, Winograd F(4x4,3x3), csrc/rz_net.hip)',
                                 'split_f16': 'k_trunk_rows on 15-row boards, else k_trunk_split (hand-written fused conv trunk: direct convolution on the f16 matrix pipe, f32 operands as hi + lo f16 pairs, f32 accumulation, csrc/rz_net.hip)',
                                 'split_f16_tiles': 'k_trunk_split (hand-written fused conv trunk: direct convolution on the f16 matrix pipe, f32 operands as hi + lo f16 pairs, f32 accumulation, csrc/rz_net.hip)',
                                 'direct': 'k_trunk (hand-written fused fp32-MFMA conv trunk, direct, csrc/rz_net.hip)'}[args.net_algo])
        elif args.evaluator == 'torchnet':
            ev = TimedEvaluator(NetEvaluator(net), torch, 'torch/MIOpen forward (~14 kernels)')
        else:
            ev = SyntheticEvaluator('vlin')
        engines.append(eng)
        evaluators.append(ev)
    evaluator = evaluators[0]
    sp = BatchedSelfPlay(engines, evaluators, temperature=TEMPERATURE, seed=0,
                         use_graph=args.graph > 0, sims_per_graph=max(args.graph, 1), eager_every=0)
    sp.warm_graphs()
    # games rank, rank+world, ... ; ids beyond the first G refill finished slots
    next_id = [rank + world * G]
    sp._start(range(G), [rank + world * i for i in range(G)])
    sp._set_active()
    finished = [0]
    first_gen_plies = []  # lengths of the finished games among the G games this rank started with
    gather_sample = []    # N > 1: finished trajectories for the one exchange of the path (after the timed region)

    def refill(n):
        ids = [next_id[0] + world * i for i in range(n)]
        next_id[0] += world * n
        return ids

    def one_step():
        # one move of every game, finished slots refilled.  --pipeline 1: the host side of a lane's move (visit counts ->
        # pi -> move, tree reuse, game step, refill) runs while the other lanes' simulations keep the GPU busy
        if args.pipeline:
            done = sp.play_move_pipelined(refill)
        else:
            done = sp.play_move()
            if done:
                free = np.nonzero(sp.slot_game < 0)[0]
                sp._start(free, refill(len(free)))
                sp.retire_finished()
        finished[0] += len(done)
        first_gen_plies.extend(len(t.moves) for t in done if t.game_id < world * G)
        if (use_dist or args.dump_trajectories) and len(gather_sample) < GATHER_SAMPLE_GAMES:
            gather_sample.extend(done[:GATHER_SAMPLE_GAMES - len(gather_sample)])

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # un-timed warm-up: the W moves asked for, and at least ~0.3 s of GPU work in all (the GPU has idled through the CPU
    # baseline and the child legs: its clocks ramp within the first moves)
    t_ramp, n_ramp = time.perf_counter(), 0
    while n_ramp < args.warmup or (time.perf_counter() - t_ramp < 0.3 and n_ramp < args.warmup + 3):
        one_step()
        n_ramp += 1
    sims0, fin0 = sp.sims_done, finished[0]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    # Kernel-level timing samples, right AFTER the timed region on the same games: every k-th graph chunk is launched
    # kernel by kernel with HIP events around the trunk, the FC GEMM and the tree step.  Not inside the timed region:
    # there the eager chunks cost 0 % (default, 80-us trunks) to 50 % (Connect4, 15-us trunks) -- the host cannot keep two
    # lanes of short kernels fed, and a stalled lane breaks the lanes' alternation for the chunks that follow
    # (profiles/r02/eager_sample_cost.txt).
    if args.eager_every > 0 and args.graph > 0:
        sims_keep, fin_keep = sp.sims_done, finished[0]
        sp.eager_every = args.eager_every
        for ev in evaluators:
            if isinstance(ev, TimedEvaluator):
                ev.record = True
        for _ in range(2):
            one_step()
        fence()
        for ev in evaluators:
            if isinstance(ev, TimedEvaluator):
                ev.record = False
        sp.eager_every = 0
        sims_sampled = sp.sims_done - sims_keep
        sp.sims_done, finished[0] = sims_keep, fin_keep
    elif args.graph <= 0:
        sims_sampled = 0
        for ev in evaluators:
            if isinstance(ev, TimedEvaluator):
                ev.record = True
        one_step()
        fence()
        for ev in evaluators:
            if isinstance(ev, TimedEvaluator):
                ev.record = False
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        counts = torch.tensor([sp.sims_done - sims0, finished[0] - fin0], dtype=torch.float64, device=red_device)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        total_sims, total_finished = float(counts[0].item()), float(counts[1].item())
    else:
        total_sims, total_finished = float(sp.sims_done - sims0), float(finished[0] - fin0)
    all_stats = sp.check()
    stats = max(all_stats, key=lambda st: st.max_slots_used)
    # the dominant kernel by itself (no other lane on the GPU): its duration on the CUs it is given
    exclusive_ms = None
    if isinstance(evaluator, TimedEvaluator) and getattr(evaluator.inner, 'hip', None) is not None and rank == 0:
        lane0 = sp.lanes[0]
        torch.cuda.synchronize()
        with torch.cuda.stream(lane0.stream):
            evs = []
            for _ in range(24):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                evaluator._trunk(lane0.eng)
                b.record()
                evs.append((a, b))
        torch.cuda.synchronize()
        exclusive_ms = sum(a.elapsed_time(b) for a, b in evs[4:]) / len(evs[4:])
    hbm_bytes = sum(st.device_bytes for st in all_stats)

    # self-play games / second (the second half of BASELINE.json's metric): keep playing, slots refilled, until
    # every game of the first generation has ended (a game ends within S plies), so the mean game length is
    # unbiased; steady-state games/s = moves/s over this leg / mean plies per game.
    selfplay = None
    if not args.no_games_leg:
        for ev in evaluators:
            if isinstance(ev, TimedEvaluator):
                ev.record = False
        fence()
        m0, t1 = sp.moves_done, time.perf_counter()
        guard = 0
        while len(first_gen_plies) < G and guard <= cells + 1:
            one_step()
            guard += 1
        fence()
        leg = time.perf_counter() - t1
        acc = [float(sp.moves_done - m0), float(sum(first_gen_plies)), float(len(first_gen_plies))]
        if use_dist:
            t = torch.tensor([leg], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            leg = float(t.item())
            c = torch.tensor(acc, dtype=torch.float64, device=red_device)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            acc = [float(v) for v in c.tolist()]
        sp.check()
        if acc[2] > 0 and leg > 0:
            mean_plies = acc[1] / acc[2]
            selfplay = {'games_per_sec': round(acc[0] / leg / mean_plies, 2), 'mean_plies_per_game': round(mean_plies, 2),
                        'games_sampled': int(acc[2]), 'moves_per_sec': round(acc[0] / leg, 2),
                        'leg_seconds': round(leg, 2),
                        'note': 'after the timed steps: all first-generation games played to their end with finished '
                                'slots refilled; games/s = moves/s of this leg / mean plies of those games'}

    # N > 1: the path's single exchange, a gather of finished trajectories to rank 0 (RCCL over xGMI when the
    # process group is nccl), exercised on a bounded sample outside the timed region
    gather = None
    merged = sorted(gather_sample, key=lambda t: t.game_id)
    if use_dist and not args.no_games_leg:
        from rlzero_amd.selfplay import gather_trajectories
        fence()
        t2 = time.perf_counter()
        try:
            # a local failure inside is agreed on by all ranks within the collectives (selfplay.gather_trajectories):
            # every rank raises together, nobody stays blocked, and the measured line is still printed
            merged = gather_trajectories(gather_sample, board, n_row, dst=0, game=args.game) or []
            gather_error = None
        except Exception as exc:  # noqa: BLE001
            merged, gather_error = [], '%s: %s' % (type(exc).__name__, exc)
        fence()
        dt = time.perf_counter() - t2
        if rank == 0 and gather_error:
            gather = {'ranks': world, 'error': gather_error[:300]}
        elif rank == 0:
            plies = sum(len(t.moves) for t in merged)
            gather = {'ranks': world, 'games': len(merged), 'plies': plies, 'backend': dist.get_backend(),
                      'payload_bytes': int(32 * len(merged) + plies * 8 * (1 + merged[0].pis.shape[1])) if merged else 0,
                      'ms': round(1000.0 * dt, 2),
                      'unique_game_ids': len({t.game_id for t in merged}) == len(merged)}

    if rank == 0 and args.dump_trajectories:
        # the finished games rank 0 holds (N > 1: gathered from all ranks), for world-size-invariance checks
        with open(args.dump_trajectories, 'w') as f:
            json.dump({str(t.game_id): {'moves': t.moves, 'winner': t.winner,
                                        'pi_hex': [float(x).hex() for x in t.pis[0]] if len(t.moves) else []}
                       for t in merged}, f)
    if rank == 0:
        value = total_sims / elapsed
        line = {
            'metric': 'mcts_sims_per_sec', 'value': round(value, 1), 'unit': 'sims/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'warmup_moves_run': n_ramp,
            'ms_per_step': round(1000.0 * elapsed / max(args.steps, 1), 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32 net (conv2/conv3 operands as hi + lo f16 pairs on the f16 MFMA pipe, f32 accumulation) / f64 tree'
                      if args.evaluator == 'hipnet' and args.net_algo.startswith('split_f16') else 'f32 net / f64 tree'), 'data': 'synthetic (random-init net, torch.manual_seed(0); '
            'games from the empty board)',
            'config': {'workload': ('connect4_6x7_n4_selfplay_%dsims_per_move_%dgames_per_gpu' % (args.playouts, G))
                       if args.game == 'connect4' else
                       'gomoku%dx%d_n%d_selfplay_%dsims_per_move_%dgames_per_gpu' % (board, board, n_row, args.playouts, G),
                       'games_total': G * world, 'c_puct': C_PUCT, 'temperature': TEMPERATURE,
                       'evaluator': args.evaluator,
                       'score_mode': 'UCT_REF (bit-exact)' if args.in_flight <= 1 else
                       'UCT_REF rule with %d simulations in flight per tree and virtual loss (opt-in, NOT the reference\'s '
                       'sequential search)' % args.in_flight,
                       'multi_sim': {'sims_in_flight': max(1, args.in_flight)},
                       'dirichlet_noise_at_every_expansion': bool(args.noise),
                       'sims_per_graph': args.graph, 'lanes': lanes, 'parallelism': 'games sharded, dp%d' % world},
            'moves_per_sec': round(total_sims / args.playouts / elapsed, 2),
            'games_finished_in_timed_region': int(total_finished),
            'selfplay_games_per_sec': selfplay['games_per_sec'] if selfplay else None,
            'selfplay': selfplay,
            'trajectory_gather': gather,
            'arena_slots_used_max': int(stats.max_slots_used),
            'engine_hbm_bytes': int(hbm_bytes),
        }
        trunk_events = [iv for ev in evaluators if isinstance(ev, TimedEvaluator) for iv in ev.events]
        if isinstance(evaluator, TimedEvaluator) and trunk_events:
            # one launch = the trunk of one lane's leaves; durations from HIP events on that lane's
            # stream.  With lanes > 1 the launches of different lanes overlap in time and share the
            # CUs, so a launch's duration is longer than the kernel needs by itself (exclusive_*).
            n_ev = len(trunk_events)
            per_stream_ms = sum(a.elapsed_time(b) for a, b in trunk_events) / n_ev
            # Launches of different lanes overlap: a lane's trunk is enqueued while the other lane's still holds the LDS
            # of the CUs, and its workgroups start CU by CU as that one drains, so the interval between a launch's two
            # events contains its wait (avg_launch_ms_per_stream; also what rocprofv3 reports per dispatch).  With
            # several lanes the duration charged to a launch is the WALL-CLOCK of the timed region / trunk launches in
            # it -- tree steps, FC GEMMs, kernel boundaries and host time all charged to the trunk: a lower bound of its
            # efficiency that needs no assumption about which intervals overlap (with the lanes' moves pipelined their
            # eager timing samples no longer coincide, so a union of sampled intervals would not mean anything).  The
            # kernel alone is exclusive_*.  With one lane this is the plain average of the event intervals (sampled on the
            # two moves behind the timed region, see above).
            launches_per_rank = lanes * (total_sims / world / G) / max(1, args.in_flight)
            ms = (elapsed * 1e3 / launches_per_rank) if lanes > 1 else per_stream_ms
            boards_per_launch = G / float(lanes) * max(1, args.in_flight)
            per_pos = trunk_flops_per_position(cells) if args.evaluator == 'hipnet' else flops_per_position(cells)
            flops = per_pos * boards_per_launch
            achieved = flops / (ms * 1e-3) / 1e12
            peak, pipe_peak, peak_note = trunk_peak(args)
            line['roofline'] = {'bound': 'mfma',
                                'kernel': '%s, %d leaves per launch' % (evaluator.label, boards_per_launch),
                                'achieved': round(achieved, 3), 'peak': round(peak, 1),
                                'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'peak_note': peak_note,
                                'traffic': pmc_traffic('k_trunk', line['config']['workload'], lanes),
                                'avg_launch_ms': round(ms, 4), 'avg_launch_ms_per_stream': round(per_stream_ms, 4),
                                'launches_timed': n_ev,
                                'note': 'achieved = ALGORITHMIC flops (direct convolution, SURVEY.md 8d) per launch / '
                                        'average launch duration. One lane: HIP events on the launch stream around every '
                                        'trunk launch of the eager sample chunks of the two moves played right behind the timed '
                                        'region (sampling inside it slows short-kernel configurations by up to 50 %). '
                                        'Several lanes: their trunk launches overlap (one is enqueued while the other '
                                        'lane\'s still holds the CUs and starts as that one drains), so avg_launch_ms = '
                                        'wall-clock of the timed region / trunk launches in it (everything else charged to the '
                                        'trunk: a lower bound), avg_launch_ms_per_stream = plain average of the event intervals '
                                        'of the eager samples (waiting included; what rocprofv3 reports per dispatch); '
                                        'exclusive_* = the same kernel launched alone after the timed region; whole_job_* = trunk flops of all '
                                        'simulations / wall-clock; mfma_executed_frac = flops the matrix pipe '
                                        'really executed / time / the peak of that pipe (split_f16: 3.35x the algorithmic '
                                        'flops on the f16 pipe; Winograd F(4x4,3x3) 3.65x fewer on the f32 pipe)',
                                'mfma_executed_frac': round(achieved / pipe_peak * executed_flop_ratio(args, cells), 4),
                                # trunk launches of all lanes x the duration charged to one / wall-clock (the eager samples
                                # run a little slower than the graph replays they stand for, so a trunk-bound run reads ~1)
                                'share_of_step_time': round(ms * lanes * (total_sims / world / G / max(1, args.in_flight)) / (elapsed * 1e3), 3),
                                'concurrent_lanes': lanes,
                                'trunk_workgroups': trunk_wgs if trunk_wgs > 0 else n_cus}
            rf = line['roofline']
            if exclusive_ms:
                ex = flops / (exclusive_ms * 1e-3) / 1e12
                rf['exclusive_launch_ms'] = round(exclusive_ms, 4)
                rf['exclusive_achieved'] = round(ex, 3)
                rf['exclusive_frac'] = round(ex / peak, 4)
                if ms > 1.3 * exclusive_ms:
                    # short kernels (small boards): in the eager samples the GPU drains its queue faster than Python
                    # refills it, and the interval between a launch's two events then contains the host's enqueue gap
                    rf['eager_samples_host_bound'] = True
            # the two small kernels of a simulation step, bracketed the same way (per stream: beside a capped trunk
            # they share 32 CUs with nothing but each other)
            fc = [x.elapsed_time(y) for ev in evaluators for x, y in ev.fc_events]
            tr = [x.elapsed_time(y) for ev in evaluators for x, y in ev.tree_events]
            if fc and tr:
                fc_ms, tr_ms = sum(fc) / len(fc), sum(tr) / len(tr)
                per_sim = tree_bytes_per_sim(365.5, 208.8, 1.74) if (args.game == 'gomoku' and board == 15) else None
                line['small_kernels'] = {'heads_gemm_ms': round(fc_ms, 4), 'tree_step_ms': round(tr_ms, 4),
                                         'launches_timed': len(tr)}
                if per_sim:
                    gbs = per_sim * boards_per_launch / (tr_ms * 1e-3) / 1e9
                    line['roofline_tree'] = {
                        'bound': 'hbm', 'kernel': 'k_tree_step_raw (expand + backup of one simulation, select of the next), '
                                                  '%d games per launch' % boards_per_launch,
                        'achieved': round(gbs, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5),
                        'traffic': pmc_traffic('k_tree_step', line['config']['workload'], lanes),
                        'avg_launch_ms': round(tr_ms, 4),
                        'note': 'achieved = the ALGORITHMIC tree bytes of the reference\'s dense formulation (SURVEY.md 8d: 12 B per '
                                'scanned child, 16 B per created child, 24 B per backed-up node, 64 B of root boards = 7.86 KB per '
                                'simulation at 15x15) x games per launch / average launch duration (HIP events); the kernel is a '
                                'chain of dependent loads per game (latency bound, one wave per game), not a streaming kernel'}
            # all trunk flops of the timed region / its whole wall-clock (tree, FC, host time included)
            whole = value / world * per_pos / 1e12
            rf['whole_job_achieved'] = round(whole, 3)
            rf['whole_job_frac'] = round(whole / peak, 4)
        else:
            per_sim = tree_bytes_per_sim(365.5, 208.8, 1.74) if board == 15 else None
            if per_sim:
                achieved = value / world * per_sim / 1e9
                line['roofline'] = {'bound': 'hbm', 'kernel': 'k_select + k_expand_backup (tree only)',
                                    'achieved': round(achieved, 3), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                    'frac': round(achieved / PEAK_HBM_GBS, 6), 'traffic': None}
        line['literal_config'] = literal
        line['configs'] = config_legs
        line['cpu_baseline'] = cpu_baseline
        print(json.dumps(line), flush=True)
    for eng in engines:
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
