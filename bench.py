#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations / second (and self-play games / second) of AlphaZero self-play on
15x15 Gomoku, 800 simulations per move, 512 games in flight per GPU = BASELINE.json configs[3] (4096 games over 8 GPUs),
random-init PolicyValueNet (torch.manual_seed(0)), f32 network / f64 tree.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one move of every game on the GPU: n_playout simulation steps (select -> evaluate -> expand / backup for all
games), pi from the root visits, a move drawn and applied, tree reuse; finished games are replaced so the batch stays
full.  The 512 games of a GPU run as four lanes of 128 (separate streams: the tree steps of one lane run beside the network
trunks of the others on the same CUs; rlzero_amd.selfplay.plan_lanes); a simulation step of a lane is two launches, trunk -> tree
step -- the reference's selection rule never reads a prior, so the policy GEMM and the priors of a search's expansions are written
in one batch per move, inside the timed region.  Games are independent: N GPUs play N x 512 games with no collective in the
timed region (weak scaling); rank 0 prints ONE JSON line.  `value` is the MEDIAN of --regions (5) timed regions of K steps
each, every region bracketed by barrier + synchronize.

On the line (what each number means and how it is priced: DESIGN.md section 5):
  roofline       the dominant kernel (the network trunk on one lane's leaves): algorithmic flops per launch / launch
                 duration, against the peak of the pipe it runs on
  roofline_tree  the tree step against the HBM roofline (algorithmic bytes of SURVEY.md 8d)
  lane_timeline  the schedule from the device-side launch trace (rlzero_amd/trace.py): CU time under trunk workgroups, trunk
                 launches on the chip, the lanes' step cycle -- no profiler in the way
  fill_1536      the same engine with 1536 games in flight (3 boards per trunk workgroup: what fills an MI355X)
  configs        the other BASELINE.json configurations (C1, C2, C3, C5) and the opt-in PUCT rule at the headline
                 geometry, each measured by a child process with its own roofline and CPU baseline
  cpu_baseline   the oracle (Python restatement of the reference, batch-1 torch CPU forward, one thread per process)
                 timed on this host's cores on a bounded sample of the same workload
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

# four lanes of games need more than HIP's default of 4 hardware queues (rlzero_amd/__init__.py does the same on import; the
# runtime reads the variable when it initialises, which is after this line in every process bench.py starts)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

BOARD, N_ROW, N_PLAYOUT, GAMES_PER_GPU, C_PUCT, TEMPERATURE = 15, 5, 800, 512, 5.0, 1.0
RESERVED_CUS_PER_XCD, N_XCD = 4, 8  # --trunk-wgs 224: CUs a capped trunk leaves to the other lane's small kernels
PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MATRIX_TFLOPS = 2500.0  # dense f16 / bf16 MFMA, same table
SPLIT_MFMAS_PER_PRODUCT = 3      # split_f16: hi*hi + hi*lo + lo*hi
PEAK_HBM_GBS = 8000.0
FILL_GAMES_PER_GPU = 1536  # three rounds of two games per CU (rounds 3-5: 2 lanes x 3 boards x 256 trunk workgroups, the batch that filled one MI355X)
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
    # split_f16_fp8 (opt-in): conv3's two cross terms as one FP8 MFMA of twice the cycles and four times the K: 2 f16-MFMA
    # equivalents per product instead of 3 on 80 % of the trunk's products (conv2 keeps 3): 0.733 of split_f16's matrix work
    mfmas = {'winograd_f4': 6030, 'direct': 21870, 'split_f16': 4050 * 16 + 270, 'split_f16_tiles': 4320 * 16 + 270,
             'split_f16_fp8': 4050 * 16 * (3 * 18432 + 2 * 73728) / (3.0 * (18432 + 73728)) + 270}[args.net_algo]
    return mfmas * 2048.0 / trunk_flops_per_position(cells)


def trunk_peak(args):
    """-> (peak TFLOP/s the trunk's ALGORITHMIC flops are priced against, peak of the pipe it executes on, note)."""
    if args.evaluator == 'hipnet' and args.net_algo == 'split_f16_fp8':
        per_product = (3 * 18432 + 2 * 73728) / float(18432 + 73728)   # 2.2 f16-MFMA equivalents per product (conv2: 3, conv3: 2)
        return (PEAK_F16_MATRIX_TFLOPS / per_product, PEAK_F16_MATRIX_TFLOPS,
                'OPT-IN arithmetic, narrower than the reference\'s f32 (RZ_NET_SPLIT_F16_FP8): peak = dense f16 MFMA peak (2500 TFLOP/s) / 2.2 '
                '-- conv3\'s cross terms hi x lo + lo x hi run as one block-scaled FP8 MFMA per tap (2 f16-MFMA equivalents per product), '
                'conv2 keeps three f16 MFMAs per product; error on the logits: tests/test_fp8_trunk.py, DESIGN.md section 5')
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


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of ``kernel`` in ``workload`` from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of that very workload, gfx950 read correction applied: profiles/r05/pmc_traffic.json, collected by
    profiles/collect_r05.sh).  None when no counter run exists for the workload."""
    for rnd in ('r06', 'r05', 'r04', 'r03'):   # (a workload without a counter run of this round keeps the last round's)
        try:
            rec = json.load(open(os.path.join(REPO, 'profiles', rnd, 'pmc_traffic.json')))
            return rec[workload]['kernels'][kernel]['traffic_bytes_per_launch']
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def rocprof_dispatch_us(kernel_substring, stats='bench_default_kernel_stats.csv'):
    """Average duration (us) and calls of the first kernel whose name contains ``kernel_substring`` in the committed rocprofv3
    --kernel-trace --stats summary of the default command (profiles/rNN/<stats>, newest round first) -> (us, calls, path) or None."""
    import csv
    for rnd in ('r06', 'r05', 'r04'):
        path = os.path.join(REPO, 'profiles', rnd, stats)
        try:
            for row in csv.DictReader(open(path)):
                if kernel_substring in row['Name']:
                    return float(row['AverageNs']) / 1e3, int(row['Calls']), os.path.relpath(path, REPO)
        except (OSError, KeyError, ValueError):
            continue
    return None


# --------------------------------------------------------------------------- power / clock while the timed regions run
class PowerSampler(object):
    """Board power while the timed regions run: the hwmon file of the GPU (sysfs; microwatts), read by a thread every 50 ms -- no child
    process, no tool.  ``mean()`` -> watts over the samples taken between start() and stop(), or None where the file does not exist."""

    def __init__(self):
        import glob
        self.paths = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') +
                            glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'))
        self.samples, self._on, self._thread = [], False, None

    def _read(self):
        best = None
        for p in self.paths:   # (one GPU is visible on the box; with several cards: the busiest)
            try:
                w = int(open(p).read().strip()) / 1e6
            except (OSError, ValueError):
                continue
            best = w if best is None or w > best else best
        return best

    def start(self):
        import threading
        if not self.paths or self._thread is not None:
            return
        self._on = True

        def loop():
            while self._on:
                w = self._read()
                if w is not None:
                    self.samples.append(w)
                time.sleep(0.05)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        self._on = False
        if self._thread is not None:
            self._thread.join(timeout=1.0)
            self._thread = None

    def mean(self):
        return round(sum(self.samples) / len(self.samples), 1) if self.samples else None


def time_resident_launches(sp, torch, n_moves=4):
    """Lone launches of the resident search (k_delta_res: one per move, every simulation of every game of the lane) between HIP
    events on the lane's own stream, each followed by the move step so that the next one searches new roots -> dict(search_ms,
    bases_ms, launches, stats) or None.  Behind the timed regions: the games it advances are played for nothing else."""
    lane = sp.lanes[0]
    eng, ev = lane.eng, getattr(lane.evaluator, 'inner', lane.evaluator)
    hip = ev.hip
    if getattr(eng, 'play_log', None) is None or not ev.resident_delta_ok(eng):
        return None
    sp.device_drain()
    pairs = []
    with sp._on(lane):
        torch.cuda.synchronize()
        hip.delta_stats(reset=True)
        for _ in range(n_moves):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            eng.flush_deferred()
            e[0].record()
            hip.delta_bases_engine(eng)                    # the bases alone (the search's own call builds them again: subtracted below)
            e[1].record()
            eng.sim_chunk(lane.evaluator, eng.n_playout)   # rz_net_search_resident: bases + ONE k_delta_res launch
            e[2].record()
            lane.uncopied.append(eng.play_move())
            pairs.append(e)
        torch.cuda.synchronize()
    stats = hip.delta_stats()
    sp.device_drain()
    bases = sum(a.elapsed_time(b) for a, b, _ in pairs) / len(pairs)
    both = sum(b.elapsed_time(c) for _, b, c in pairs) / len(pairs)
    return {'search_ms': both - bases, 'bases_ms': bases, 'launches': len(pairs), 'stats': stats}


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
            'sample': '%d processes x %.0f s of %s at %d sims/move, %s + batch-1 torch CPU forward, 1 thread each'
                      % (cores, seconds, what, n_playout, 'oracle/muzero_ref.py' if game == 'muzero' else 'oracle/mcts_ref.py')}


def child_line(flags, timeout=900):
    """Run `bench.py <flags>` as a child process (this process has not touched the GPU) -> its JSON line or None."""
    cmd = [sys.executable, os.path.abspath(__file__)] + [str(f) for f in flags]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=REPO, timeout=timeout).stdout
        return json.loads([ln for ln in out.decode().splitlines() if ln.startswith('{')][-1])
    except Exception:  # noqa: BLE001
        return None


def run_fill_config(args):
    """The same engine with 1536 games in flight (the layout plan_lanes picks: with the receptive-field trunk ONE resident lane whose
    launch runs in three rounds of two games per CU; before, two lanes of 768), as a child process -> the fields of its line worth keeping."""
    rec = child_line(['--games', FILL_GAMES_PER_GPU, '--steps', args.steps, '--warmup', max(args.warmup, 3),
                      '--regions', 1, '--net-algo', args.net_algo, '--graph', args.graph, '--noise', args.noise, '--deferred', args.deferred,
                      '--no-cpu-baseline', '--no-games-leg', '--no-fill', '--no-configs', '--timeline', 0], 600)
    if rec is None:
        return None
    rf = rec.get('roofline') or {}
    return {'workload': rec['config']['workload'], 'value': rec['value'], 'ms_per_step': rec['ms_per_step'], 'lanes': rec['config'].get('lanes'),
            'frac': rf.get('frac'), 'avg_launch_ms': rf.get('avg_launch_ms'), 'traffic': rf.get('traffic')}


# the other configurations of BASELINE.json, each measured by a child process of the default N = 1 run
CONFIG_LEGS = (  # (key, flags, seconds of CPU baseline at --cpu-seconds 60); a leg starts on a GPU that has idled: warm-up moves
    # (a TicTacToe move is a third of a millisecond: regions of 180 moves = 20+ whole games per slot, so that the fence of a region --
    # synchronize, then the host reads the last moves' log rows -- is not most of what is timed; finished slots refill on the device)
    ('C1_ttt_25sims_1game', ['--board', 3, '--playouts', 25, '--games', 1, '--lanes', 1, '--steps', 180, '--warmup', 20], 5.0),
    ('C1_16games', ['--board', 3, '--playouts', 25, '--games', 16, '--lanes', 1, '--steps', 180, '--warmup', 20, '--no-cpu-baseline'], 0.0),
    ('C2_9x9_200sims_64games', ['--board', 9, '--playouts', 200, '--games', 64, '--lanes', 1, '--steps', 32, '--warmup', 8], 12.0),
    ('C2_16_in_flight', ['--board', 9, '--playouts', 200, '--games', 64, '--steps', 8, '--warmup', 8, '--in-flight', 16,   # (1024 leaves: two lanes)
                         '--no-cpu-baseline'], 0.0),
    ('C3_connect4_400sims_512games', ['--game', 'connect4', '--playouts', 400, '--games', 512, '--steps', 6, '--warmup', 6], 12.0),
    ('C4_puct_rule', ['--score-mode', 'puct', '--steps', 3, '--warmup', 2, '--no-cpu-baseline'], 0.0),
    # OPT-IN arithmetic (RZ_NET_SPLIT_F16_FP8), never the headline: what the 1e-4 of the path is worth on this chip
    ('C4_optin_fp8_cross_terms', ['--net-algo', 'split_f16_fp8', '--steps', 3, '--warmup', 3, '--no-cpu-baseline'], 0.0),
    # (launches of 16 moves, two kept enqueued ahead of the host: the timed region ends with the host reading the last two -- 2048
    # moves = 128 launches = 0.6 s keep that drain at 2 % of the region)
    ('C5_muzero_cartpole_50sims_8192envs', ['--game', 'muzero', '--playouts', 50, '--games', 8192, '--steps', 2048, '--warmup', 256], 12.0),
)


def run_timeline_leg(args):
    """The device-side launch trace of the same layout (rlzero_amd/trace.py) in a child process of its own: a diagnostic that
    builds a second self-play object with traced kernel instantiations must not be able to take the measured line with it."""
    rec = child_line(['--timeline-leg', '--games', args.games, '--playouts', args.playouts, '--lanes', args.lanes, '--noise', args.noise,
                      '--device-moves', args.device_moves], 300)
    return rec if rec is not None else {'error': 'the child process printed no line'}


def run_config_legs(args):
    out = {}
    for key, flags, cpu_s in CONFIG_LEGS:
        rec = child_line(flags + ['--cpu-seconds', max(1.0, cpu_s * args.cpu_seconds / 60.0), '--regions', 1, '--no-games-leg',
                                  '--no-fill', '--no-configs', '--timeline', 0] + (['--no-cpu-baseline'] if args.no_cpu_baseline else []), 600)
        if rec is None:
            out[key] = {'error': 'the child process printed no line'}
            continue
        rf = rec.get('roofline') or {}
        cpu = rec.get('cpu_baseline') or {}
        leg = {'workload': rec['config']['workload'], 'value': rec['value'], 'ms_per_step': rec['ms_per_step'],
               'roofline': {k: rf.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms')},
               'cpu_baseline': {k: cpu.get(k) for k in ('value', 'cores')} if cpu else None}
        rt = rec.get('roofline_tree')
        if rt:
            leg['roofline_tree'] = {k: rt.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms')}
        out[key] = leg
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
def delta_resident_per_cu(args):
    """Resident workgroups a CU holds on the run's route: two of the receptive-field kernel (k_delta_res: the default trunk on boards of
    11 .. 16 rows and columns, unless RZ_NET_DELTA / RZ_NET_DELTA_RESIDENT / RZ_RESIDENT switch it off), one otherwise."""
    if getattr(args, 'net_algo', 'split_f16') != 'split_f16' or os.environ.get('RZ_RESIDENT') == '0':
        return 1
    rows, cols = (6, 7) if getattr(args, 'game', 'gomoku') == 'connect4' else (args.board, args.board)
    if 11 <= rows <= 16:
        return 2 if (os.environ.get('RZ_NET_DELTA', '1') != '0' and os.environ.get('RZ_NET_DELTA_RESIDENT', '1') != '0') else 1
    from rlzero_amd.engine import compact_grid_board   # (boards of the compact LDS grid: k_trunk_split<RES> holds two games per CU as well)
    return 2 if compact_grid_board(rows, cols) else 1


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
    def hip(self):
        return self.inner.hip

    def deferred_ok(self, eng):
        ok = getattr(self.inner, 'deferred_ok', None)
        return ok is not None and ok(eng)

    def resident_ok(self, eng):
        ok = getattr(self.inner, 'resident_ok', None)
        return ok is not None and ok(eng)

    def search_resident(self, eng, n_sims, select_first=False):
        return self.inner.search_resident(eng, n_sims, select_first)

    def prepare_search(self, eng):
        prep = getattr(self.inner, 'prepare_search', None)
        if prep is not None:
            prep(eng)

    def deferred_trunk(self, eng):
        """Deferred-priors route (two launches per step): the trunk bracketed when recording; the event behind it and the
        first event of the NEXT step of the same eager chunk bracket the tree step launched in between."""
        if not self.record:
            self.last_c = None
            return self.inner.deferred_trunk(eng)
        t = self.torch
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        a.record()
        if getattr(self, 'last_c', None) is not None:
            self.tree_events.append((self.last_c, a))
        out = self.inner.deferred_trunk(eng)
        b.record()
        self.events.append((a, b))
        self.last_c = b
        return out

    @property
    def fused_heads(self):
        return getattr(self.inner, 'fused_heads', False)

    def raw_heads(self, eng):
        """Fused route (the tree kernel finishes the heads): k_trunk bracketed when recording; a third event behind
        the FC GEMM brackets that kernel and, with the first event of the NEXT simulation of the same eager chunk,
        the tree step launched in between (same stream)."""
        if not self.record:
            self.last_c = None
            if os.environ.get('RZ_DIAG_SKIP_FC') == '1':
                # DIAGNOSTIC ONLY (wrong trees): the FC GEMM is launched once and its stale outputs reused, so a lane's chain is
                # trunk -> tree step -- what a shorter chain would buy the schedule (profiles/r04/NOTES.md)
                self._trunk(eng)
                if getattr(self, '_stale_heads', None) is None:
                    self._stale_heads = self.inner.hip.heads_gemm(eng.obs.shape[0])
                return self._stale_heads
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
    workload = 'muzero_cartpole_v1_%dsims_per_move_%denvs_per_gpu' % (n_sims, G)
    torch.manual_seed(0)
    net = MuZeroNet().to(device).eval()
    sp = MuZeroSelfPlay(net, CartPoleBatch(G, device, seed=rank), n_sims=n_sims, seed=rank, fused=bool(args.mz_fused),
                        moves_per_launch=args.mz_moves_per_launch)
    if args.mz_gpw:
        sp.tree.set_search_shape(args.mz_gpw)
    sp.collect(args.warmup)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # (the interpreter's own housekeeping out of the region: a full collection now, and the objects alive so far -- torch's modules,
    # the import graph -- moved out of the collector's sight; one generation-2 pass inside the region stalled the host for 38 ms,
    # four launches of the GPU)
    import gc
    gc.collect()
    gc.freeze()
    sims0 = sp.sims_done
    sp.sim_events = []  # HIP events around every 8th simulation step (one hipGraph replay) of the timed region
    if sp.fused:
        sp.search_events = []  # ... or around every fused search launch (all simulations of a move)
    t0 = time.perf_counter()
    # (fused moves: launches of sp.moves_per_launch moves, records read two launches behind.  The episodes stay alive until the clock
    # has stopped: dropping them -- 10 MB of records per launch, 1.3 GB over 2048 moves -- inside the region cost 57 ms of munmap)
    episodes = sp.collect(args.steps)
    finished = len(episodes)
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
    del episodes
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
                # whole MOVES: the four 64 x 64 layers run on the f16 matrix pipe with hi + lo operand pairs (three MFMAs per f32
                # product): priced like the headline's trunk, against dense f16 / 3; the f32-MFMA figure stays as a second key
                # (the move-by-move route, fused_moves off, runs its layers on the f32-input MFMA)
                f16_layers = bool(sp.fused_moves)
                peak = PEAK_F16_MATRIX_TFLOPS / SPLIT_MFMAS_PER_PRODUCT if f16_layers else PEAK_FP32_MATRIX_TFLOPS
                roofline = {'bound': 'mfma', 'kernel': sp.sim_step_label + ', %d environments per launch' % G,
                            'achieved': round(tf, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                            'frac': round(tf / peak, 5), 'frac_of_f32_mfma_peak': round(tf / PEAK_FP32_MATRIX_TFLOPS, 5),
                            'traffic': pmc_traffic('k_mz_search', workload), 'avg_launch_ms': round(ms, 4),
                            'launches_timed': len(events), 'hbm_achieved_gbs': round(gbs, 2),
                            'moves_per_launch': moves_per_launch,
                            # the launches of the timed region back to back on the GPU (first launch's start to last one's end) against
                            # the region's wall clock: the difference is the host reading the last launches' episodes behind the GPU
                            'gpu_span_ms': round(events[0][0].elapsed_time(events[-1][1]), 2), 'region_wall_ms': round(1e3 * elapsed, 2),
                            'gpu_gaps_ms': [round(g, 2) for g in sorted(events[i][1].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1))[-4:]],
                            'note': 'latency-bound: one wave per workgroup walks 16 trees between the layer stages (DESIGN.md section 4)'}
            else:
                roofline = {'bound': 'hbm', 'kernel': sp.sim_step_label + ', %d environments per launch' % G,
                            'achieved': round(gbs, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5),
                            'traffic': None, 'avg_launch_ms': round(ms, 4), 'launches_timed': len(events),
                            'flops_achieved_tflops': round(tf, 3), 'flops_frac_of_f32_mfma_peak': round(tf / PEAK_FP32_MATRIX_TFLOPS, 5)}
        print(json.dumps({
            'metric': 'mcts_sims_per_sec', 'value': round(total / elapsed, 1), 'unit': 'sims/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000.0 * elapsed / max(args.steps, 1), 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 model / f64 tree',
            'data': 'synthetic (random-init MuZero MLPs, torch.manual_seed(0); CartPole-v1 restatement)',
            'config': {'workload': workload,
                       'games_total': G * world, 'discount': 0.997, 'parallelism': 'environments sharded, dp%d' % world},
            'episodes_finished_in_timed_region': finished, 'tree_hbm_bytes': int(sp.tree.device_bytes),
            'roofline': roofline, 'cpu_baseline': cpu_baseline}), flush=True)
    sp.close()
    if use_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)   # (moves per timed region: a 22-ms move makes five regions of 20 a 2-s measurement; with 4 the
    ap.add_argument('--warmup', type=int, default=5)   # fence at a region's end -- synchronize, the host reads the last log rows -- is 3 % of it)
    ap.add_argument('--cpu-worker', type=float, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-seconds', type=float, default=60.0,
                    help='wall-clock budget of the CPU baseline of the main workload (SURVEY.md 8d: >= 60 s); the '
                         'baselines of the `configs` legs get 5-12 s each, scaled with this value')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the child-process legs for the other BASELINE.json configurations (C1, C2, C3, C5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--games', type=int, default=0,
                    help='games per GPU; 0 = %d (BASELINE.json configs[3]: 4096 games / 8 GPUs)' % GAMES_PER_GPU)
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
    ap.add_argument('--graph', type=int, default=16, help='simulation steps per hipGraph (0 = eager); 16 or more: +3 %% over 8 on four lanes')
    ap.add_argument('--net-algo', default='split_f16', choices=['winograd_f4', 'direct', 'split_f16', 'split_f16_tiles', 'split_f16_fp8'],
                    help="'split_f16_fp8': OPT-IN arithmetic narrower than the reference's f32 (never the default; 15x15 Gomoku only)")
    ap.add_argument('--no-games-leg', action='store_true',
                    help='skip the self-play games/s leg (after the timed steps the games of the first generation '
                         'are played to their end, slots refilled, to measure moves/s over whole games and the mean '
                         'game length)')
    ap.add_argument('--no-fill', '--no-literal-config', dest='no_fill', action='store_true',
                    help='skip the extra N=1 measurement with %d games in flight (child process)' % FILL_GAMES_PER_GPU)
    ap.add_argument('--regions', type=int, default=5, help='timed regions of K steps each; value = their median')
    ap.add_argument('--score-mode', default='uct_ref', choices=['uct_ref', 'puct'],
                    help="uct_ref: the reference's selection rule (node.py:32-42, 75-88; bit-exact); puct: the opt-in AlphaZero "
                         "rule Q + c P sqrt(N_parent) / (N + 1) (every level scans all children; parity by the oracle's restatement)")
    ap.add_argument('--heads-algo', default='auto', choices=['auto', 'f32', 'split32', 'split64', 'parts', 'in_trunk'],
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
    ap.add_argument('--device-moves', type=int, default=1,
                    help='1 (default): the move step on the device (rz_play_*: the draw, tree reuse, game step, end / refill of slots as '
                         'kernels enqueued moves ahead of the host, which reads the log behind the GPU, forms pi with numpy and verifies '
                         'every move); 0: the host-driven move step of rounds 1-4 (--pipeline)')
    ap.add_argument('--timeline-leg', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--mz-moves-per-launch', type=int, default=16, help='MuZero: moves of every environment per launch of the whole-moves kernel')
    ap.add_argument('--mz-gpw', type=int, default=0,
                    help='--game muzero: games per workgroup of k_mz_search (rz_mz_set_search_shape; 0 = automatic)')
    ap.add_argument('--mz-fused', type=int, default=1,
                    help='--game muzero: 1 = the whole search of a move in one kernel launch (k_mz_search), 0 = one hipGraph '
                         'of tree kernels + PyTorch-ROCm layers per simulation')
    ap.add_argument('--in-flight', type=int, default=1,
                    help='K > 1: opt-in virtual-loss mode, K simulations of every tree share one evaluator batch (NOT the '
                         'reference\'s sequential search: results differ from it; for batches too small to fill the GPU)')
    ap.add_argument('--deferred', type=int, default=1,
                    help='1 (default): deferred priors where the route exists (UCT_REF, one simulation in flight, boards of 11 .. 16 rows) -- '
                         'a step is trunk -> tree step, the policy GEMM and the priors of a move\'s expansions run as one batch per move; '
                         '0: the three-launch step (trunk -> FC GEMM -> tree step writing the priors at once)')
    ap.add_argument('--timeline', type=int, default=1,
                    help='1: behind the timed work, play three more moves of the same layout with the device-side launch trace '
                         'attached (rlzero_amd/trace.py) and report the schedule (`lane_timeline`, roofline.launches_in_flight)')
    ap.add_argument('--lanes', type=int, default=0,
                    help='independent batches of games on separate HIP streams (the tree / FC kernels of one lane run beside the '
                         'network trunks of the others); 0 = what rlzero_amd.selfplay.plan_lanes picks for the batch (512 games: 4)')
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

    if args.timeline_leg:   # child mode: the launch trace of the layout, one JSON line
        import torch
        from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
        from rlzero_amd.selfplay import plan_lanes
        from rlzero_amd.trace import measure
        torch.manual_seed(0)
        G = args.games if args.games > 0 else GAMES_PER_GPU
        net = PolicyValueNet(BOARD).to('cuda:0').eval()
        n_cus = torch.cuda.get_device_properties(0).multi_processor_count
        lanes = args.lanes if args.lanes > 0 else plan_lanes(G, n_cus, deferred=True, cells=BOARD * BOARD, resident_per_cu=delta_resident_per_cu(args))[0]
        print(json.dumps(measure(net, BOARD, N_ROW, n_games=G, n_playout=args.playouts, lanes=lanes, device='cuda:0',
                                 add_noise=bool(args.noise), device_moves=bool(args.device_moves), resident_search=False)), flush=True)
        return

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU
    cpu_baseline = None
    default_config = (args.game == 'gomoku' and args.board == BOARD and args.playouts == N_PLAYOUT)
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        playouts = 50 if (args.game == 'muzero' and args.playouts == N_PLAYOUT) else args.playouts
        cpu_baseline = run_cpu_baseline(args.cpu_seconds, args.game, args.board, playouts)

    # the same engine with 1536 games in flight, in a child process of its own, also before this process touches the GPU
    # (a process that has initialised the GPU starts no program)
    fill = None
    if (world == 1 and args.gpus == 1 and not args.no_fill and default_config and args.games == 0
            and args.evaluator == 'hipnet' and args.score_mode == 'uct_ref'):
        fill = run_fill_config(args)
    timeline = None
    if (world == 1 and args.gpus == 1 and args.timeline and default_config and args.evaluator == 'hipnet' and args.score_mode == 'uct_ref'
            and args.net_algo.startswith('split_f16') and args.deferred and args.in_flight <= 1 and (args.games == 0 or args.games > 256)):
        timeline = run_timeline_leg(args)
    # ... and the other configurations of BASELINE.json (C1, C2, C3, C5), each a child process with its own
    # roofline and CPU baseline: reported under `configs` of the default N = 1 line
    config_legs = None
    if (world == 1 and args.gpus == 1 and not args.no_configs and default_config and args.games == 0
            and args.evaluator == 'hipnet' and args.score_mode == 'uct_ref'):
        config_legs = run_config_legs(args)

    # one process per GPU: this rank's host thread onto the cores of ITS GPU's NUMA node (sysfs only, no numactl / taskset hop, before
    # the first GPU call; after the CPU baseline and the child legs, which use every core) -- rlzero_amd/affinity.py
    from rlzero_amd.affinity import pin_to_gpu
    one_device = os.environ.get('RZ_BENCH_SINGLE_DEVICE') == '1'
    host_affinity = pin_to_gpu(0 if one_device else local_rank, 1 if one_device else int(os.environ.get('LOCAL_WORLD_SIZE', world)),
                               apply=os.environ.get('RZ_BENCH_NO_PIN') != '1')

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

    affinities = [host_affinity]
    if use_dist:
        affinities = [None] * world
        dist.all_gather_object(affinities, host_affinity)
    if args.game == 'muzero':
        run_muzero(args, rank, world, device, dist, red_device, use_dist, cpu_baseline)
        return
    board, n_row = args.board, (N_ROW if args.board >= 5 else args.board)
    cells = board * board
    if args.game == 'connect4':
        board, n_row, cells = (6, 7), 4, 42
    n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    if args.lanes > 0:
        lanes = args.lanes
    else:
        from rlzero_amd.selfplay import plan_lanes
        will_defer = (bool(args.deferred) and args.evaluator == 'hipnet' and args.net_algo in ('split_f16', 'split_f16_fp8') and args.game == 'gomoku'
                      and 11 <= args.board <= 16 and args.score_mode == 'uct_ref' and args.in_flight <= 1)
        small_trunk = (bool(args.deferred) and args.evaluator == 'hipnet' and args.net_algo in ('split_f16', 'split_f16_tiles')
                       and args.score_mode == 'uct_ref' and args.in_flight <= 1)
        lanes = plan_lanes((args.games if args.games > 0 else GAMES_PER_GPU) * max(1, args.in_flight), n_cus, deferred=will_defer,
                           cells=cells if (small_trunk or args.in_flight > 1) else None, in_flight=max(1, args.in_flight),
                           resident_per_cu=delta_resident_per_cu(args) if (will_defer or small_trunk) else 1)[0]
    trunk_wgs = max(0, args.trunk_wgs)
    # default batch: the 512 games per GPU of BASELINE.json configs[3] (4096 games over 8 GPUs), as four lanes of 128
    G = args.games if args.games > 0 else GAMES_PER_GPU
    heads_algo = args.heads_algo
    if heads_algo == 'auto' and lanes > 1 and trunk_wgs == 0 and args.evaluator == 'hipnet' and args.net_algo.startswith('split_f16'):
        heads_algo = 'parts'  # un-capped lanes: the LDS-free GEMM that fits beside a resident trunk workgroup
        from rlzero_amd.selfplay import fc_in_trunk_pays
        if fc_in_trunk_pays(*((6, 7, 7) if args.game == 'connect4' else (board, board, board * board))):
            heads_algo = 'in_trunk'  # small FC layers: the trunk's workgroups run them on their own boards, no GEMM launch
    lanes = max(1, min(lanes, G))
    per_lane = [G // lanes + (1 if i < G % lanes else 0) for i in range(lanes)]
    torch.manual_seed(0)  # identical weights on every rank
    net = (PolicyValueNet(6, 7, 7) if args.game == 'connect4' else PolicyValueNet(board)).to(device).eval()
    net_shape = (6, 7, 7) if args.game == 'connect4' else board
    engines, evaluators = [], []
    deferred_route = resident_route = False
    for g_lane in per_lane:
        eng = MCTSEngine(board, n_row, n_games=g_lane, n_playout=args.playouts, c_puct=C_PUCT, device=device,
                         game=args.game, add_noise=bool(args.noise), noise_seed=1000 * rank + len(engines),
                         sims_in_flight=args.in_flight, score_mode=args.score_mode)
        if args.evaluator == 'hipnet':
            hip_ev = HipNetEvaluator(net, net_shape, device, max_boards=eng.n_leaves)
            hip_ev.hip.set_algo(args.net_algo)
            hip_ev.hip.set_heads_algo(heads_algo)
            hip_ev.hip.set_max_workgroups(trunk_wgs)
            hip_ev.deferred_priors = bool(args.deferred)
            if os.environ.get('RZ_RESIDENT') == '0':   # (profiles/ab_resident.sh: the two-launch step on a batch the resident search would take)
                hip_ev.resident_search = False
            deferred_route = hip_ev.deferred_ok(eng)
            resident_route = (lanes == 1 or G <= n_cus * hip_ev.resident_per_cu(eng)) and hip_ev.resident_ok(eng)   # (BatchedSelfPlay switches it off for lanes that share CUs)
            ev = TimedEvaluator(hip_ev, torch,
                                {'winograd_f4': 'k_trunk_wino_f4<4>',
                                 'split_f16': 'k_trunk_rows' if (args.game == 'gomoku' and 11 <= board <= 16) else 'k_trunk_split',
                                 'split_f16_tiles': 'k_trunk_split', 'split_f16_fp8': 'k_trunk_rows',
                                 'direct': 'k_trunk'}[args.net_algo])
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
    device_moves = bool(args.device_moves) and args.in_flight <= 1
    if device_moves:   # the slots take their games from a queue on the device, finished ones refill themselves
        sp.device_attach(queue_capacity=1 << 16)
        sp.device_queue([rank + world * i for i in range(1 << 16)])
    else:
        sp._start(range(G), [rank + world * i for i in range(G)])
        sp._set_active()
    finished = [0]
    first_gen_plies = []  # lengths of the finished games among the G games this rank started with
    gather_sample = []    # N > 1: finished trajectories for the one exchange of the path (after the timed region)

    def refill(n):
        ids = [next_id[0] + world * i for i in range(n)]
        next_id[0] += world * n
        return ids

    def account(done):
        finished[0] += len(done)
        first_gen_plies.extend(len(t.moves) for t in done if t.game_id < world * G)
        if (use_dist or args.dump_trajectories) and len(gather_sample) < GATHER_SAMPLE_GAMES:
            gather_sample.extend(done[:GATHER_SAMPLE_GAMES - len(gather_sample)])

    def one_step():
        # one move of every game, finished slots refilled.  --device-moves 1: the whole move is enqueued (search, draw, priors, tree
        # reuse, game step, refill); the host reads the log a few moves behind (pi by numpy, every move verified).
        # --device-moves 0 --pipeline 1: the host side of a lane's move (visit counts ->
        # pi -> move, tree reuse, game step, refill) runs while the other lanes' simulations keep the GPU busy
        if device_moves:
            return account(sp.play_move_device())
        if args.pipeline:
            done = sp.play_move_pipelined(refill)
        else:
            done = sp.play_move()
            if done:
                free = np.nonzero(sp.slot_game < 0)[0]
                sp._start(free, refill(len(free)))
                sp.retire_finished()
        account(done)

    def fence():
        torch.cuda.synchronize()
        if device_moves:   # every enqueued move has ended: read the rows still unread (pi, verification, finished games)
            account(sp.device_drain())
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # un-timed warm-up: the W moves asked for, and at least ~0.3 s of GPU work in all (the GPU has idled through the CPU
    # baseline and the child legs: its clocks ramp within the first moves)
    t_ramp, n_ramp = time.perf_counter(), 0
    while n_ramp < args.warmup or (time.perf_counter() - t_ramp < 0.3 and n_ramp < args.warmup + 3):
        one_step()
        n_ramp += 1
    # (the interpreter's housekeeping out of the regions, as in run_muzero: a generation-2 pass over torch's object graph stalls the
    # host for tens of milliseconds)
    import gc
    gc.collect()
    if os.environ.get('RZ_BENCH_NO_GC_FREEZE') != '1':   # (A / B: profiles/r04/NOTES.md)
        gc.freeze()
    # --regions timed regions of K steps each, every one bracketed by barrier + synchronize; the MEDIAN region is reported
    regions = []
    power = PowerSampler()
    for _ in range(max(1, args.regions)):
        fence()   # (device-driven moves: also reads the rows of the moves before the region -- counted before it, not in it)
        sims0, fin0 = sp.sims_done, finished[0]
        power.start()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        fence()
        dt = time.perf_counter() - t0
        power.stop()
        if use_dist:
            # every rank's own (seconds, simulations, games finished) of the region, on every rank: the line reports MAX time /
            # SUM work (the contract) AND each rank's own rate, so a throttling or straggling GPU shows
            mine = torch.tensor([dt, sp.sims_done - sims0, finished[0] - fin0], dtype=torch.float64, device=red_device)
            rows = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
            rows = torch.stack(rows).cpu().numpy()
            regions.append((float(rows[:, 0].max()), float(rows[:, 1].sum()), float(rows[:, 2].sum()),
                            [float(r[1] / r[0]) for r in rows]))
        else:
            regions.append((dt, float(sp.sims_done - sims0), float(finished[0] - fin0), [float(sp.sims_done - sims0) / dt]))
    elapsed, total_sims, total_finished, per_rank_rates = sorted(regions, key=lambda r: r[1] / r[0])[len(regions) // 2]
    # the resident search's kernel by itself: lone launches between HIP events on its own stream, behind the timed regions
    resident_timing = None
    if rank == 0 and resident_route and device_moves and len(sp.lanes) == 1:
        sims_keep, fin_keep = sp.sims_done, finished[0]
        try:
            resident_timing = time_resident_launches(sp, torch)
        except Exception as exc:  # noqa: BLE001  (a diagnostic cannot take the line with it)
            resident_timing = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:200])}
        sp.sims_done, finished[0] = sims_keep, fin_keep
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
    all_stats = sp.check()
    stats = max(all_stats, key=lambda st: st.max_slots_used)
    reuse_dropped_timed = int(sum(st.reuse_dropped for st in all_stats))   # kept subtrees over the carry limit (0 = the reference's unbounded update_with_move)
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
        leg_stats = sp.check()
        dropped = [float(sum(st.reuse_dropped for st in leg_stats)), float(max(st.max_slots_used for st in leg_stats))]
        if use_dist:
            c = torch.tensor(dropped, dtype=torch.float64, device=red_device)
            dist.all_reduce(c, op=dist.ReduceOp.MAX)
            dropped = [float(v) for v in c.tolist()]
        if acc[2] > 0 and leg > 0:
            mean_plies = acc[1] / acc[2]
            selfplay = {'games_per_sec': round(acc[0] / leg / mean_plies, 2), 'mean_plies_per_game': round(mean_plies, 2),
                        'games_sampled': int(acc[2]), 'moves_per_sec': round(acc[0] / leg, 2),
                        'leg_seconds': round(leg, 2),
                        # whole games with refills: subtrees dropped at the carry limit since the engines were created (max over
                        # ranks; 0 = every update_with_move kept its subtree, alphazero_mcts.py:96-103) and the fullest arena
                        'reuse_dropped': int(dropped[0]), 'arena_slots_used_max': int(dropped[1]),
                        'arena_slots': int(leg_stats[0].arena_slots)}

    # N > 1: the path's single exchange, a gather of finished trajectories to rank 0 (RCCL over xGMI when the
    # process group is nccl), exercised on a bounded sample outside the timed region
    gather = None
    merged = sorted(gather_sample, key=lambda t: t.game_id)
    if use_dist and not args.no_games_leg:
        from rlzero_amd.selfplay import COLLECTIVES_PER_EXCHANGE, gather_trajectories
        fence()
        t2 = time.perf_counter()
        try:
            # a local failure inside is agreed on by all ranks within the collectives (selfplay.gather_trajectories):
            # every rank raises together, nobody stays blocked, and the measured line is still printed
            # (pi as float32: what the learner consumes, half the bytes -- the exchange SURVEY.md 8e sizes)
            merged = gather_trajectories(gather_sample, board, n_row, dst=0, game=args.game, pi_dtype=np.float32) or []
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
                      'payload_bytes': int(32 * len(merged) + plies * (8 + 4 * merged[0].pis.shape[1])) if merged else 0,
                      'ms': round(1000.0 * dt, 2),
                      'collectives_per_exchange': COLLECTIVES_PER_EXCHANGE,   # one size all_gather + one payload gather
                      'unique_game_ids': len({t.game_id for t in merged}) == len(merged)}

    if rank == 0 and args.dump_trajectories:
        # the finished games rank 0 holds (N > 1: gathered from all ranks), for world-size-invariance checks
        with open(args.dump_trajectories, 'w') as f:
            json.dump({str(t.game_id): {'moves': t.moves, 'winner': t.winner,
                                        'pi_hex': [float(np.float32(x)).hex() for x in t.pis[0]] if len(t.moves) else []}
                       for t in merged}, f)
    if rank == 0:
        value = total_sims / elapsed
        split = args.evaluator == 'hipnet' and args.net_algo.startswith('split_f16')
        line = {
            'metric': 'mcts_sims_per_sec', 'value': round(value, 1), 'unit': 'sims/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000.0 * elapsed / max(args.steps, 1), 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('OPT-IN: f32 net with conv3\'s cross terms on 8-bit operands (hi x hi on f16, hi x lo + lo x hi on the block-scaled FP8 pipe, '
                      'f32 accumulation) / f64 tree' if args.net_algo == 'split_f16_fp8' else
                      'f32 net (hi + lo f16 operand pairs on the f16 MFMA pipe, f32 accumulation) / f64 tree') if split else 'f32 net / f64 tree',
            'data': 'synthetic (random-init net, torch.manual_seed(0); games from the empty board)',
            'config': {'workload': ('connect4_6x7_n4_selfplay_%dsims_per_move_%dgames_per_gpu' % (args.playouts, G))
                       if args.game == 'connect4' else
                       'gomoku%dx%d_n%d_selfplay_%dsims_per_move_%dgames_per_gpu' % (board, board, n_row, args.playouts, G),
                       'games_total': G * world, 'lanes': lanes, 'c_puct': C_PUCT, 'temperature': TEMPERATURE,
                       'evaluator': args.evaluator,
                       'score_mode': ('UCT_REF (bit-exact)' if args.score_mode == 'uct_ref' else 'PUCT (opt-in)') +
                       ('' if args.in_flight <= 1 else ', %d simulations in flight per tree (opt-in virtual loss)' % args.in_flight),
                       'multi_sim': {'sims_in_flight': max(1, args.in_flight)},
                       'dirichlet_noise': bool(args.noise), 'sims_per_graph': args.graph,
                       'priors': 'deferred (one policy GEMM + one priors kernel per move, outside the chain of a simulation step)'
                       if deferred_route else 'written by every tree step',
                       'launches_per_step': (0 if resident_route else 2) if deferred_route else 3,
                       'resident_search': bool(resident_route),
                       'move_step': 'device (rz_play_*: moves enqueued ahead, log read behind; pi and verification by numpy on the host)'
                       if device_moves else ('host, pipelined under the other lanes' if args.pipeline else 'host'),
                       'hw_queues': int(os.environ.get('GPU_MAX_HW_QUEUES', '4')),   # hardware queues asked of the HIP runtime (a lane each)
                       'parallelism': 'games sharded, dp%d' % world},
            'regions_sims_per_sec': [round(r[1] / r[0], 1) for r in regions], 'warmup_moves_run': n_ramp,
            # every rank's own rate in the reported (median) region -- its simulations / ITS seconds; value = SUM work / MAX seconds
            'per_rank_sims_per_sec': [round(v, 1) for v in per_rank_rates],
            'per_rank_min': round(min(per_rank_rates), 1), 'per_rank_max': round(max(per_rank_rates), 1),
            'moves_per_sec': round(total_sims / args.playouts / elapsed, 2),
            'games_finished_in_timed_region': int(total_finished),
            'selfplay_games_per_sec': selfplay['games_per_sec'] if selfplay else None,
            'selfplay': selfplay,
            'trajectory_gather': gather,
            'host_affinity': affinities,   # per rank: the NUMA node of its GPU and the cores its host thread is pinned to

            'arena_slots_used_max': int(stats.max_slots_used), 'arena_slots': int(stats.arena_slots),
            'reuse_dropped': reuse_dropped_timed,   # rank 0, warm-up + timed regions + timing samples
            # device-driven moves: draws the device left to the host (u within 1e-10 of an interval edge); every other move verified
            'move_draws_left_to_host': int(getattr(sp, 'stalls_resolved', 0)) if device_moves else None,
            'engine_hbm_bytes': int(hbm_bytes),
            # the receptive-field trunk's cache of the roots' activations (rz_net_delta_reserve: a header + 2 x 104 448 bytes per game)
            'base_cache_bytes': int(sum(getattr(getattr(getattr(ev, 'inner', ev), 'hip', None), '_delta_games', 0) * (2 * 104448 + 80) for ev in evaluators)),
        }
        trunk_events = [iv for ev in evaluators if isinstance(ev, TimedEvaluator) for iv in ev.events]
        if isinstance(evaluator, TimedEvaluator) and trunk_events:
            # One launch = the trunk on one lane's leaves.  With several lanes a launch is enqueued while the other lane's
            # trunk still holds the CUs, so the interval between its two events contains its wait (avg_launch_ms_per_stream,
            # also what rocprofv3 reports per dispatch): the duration charged to a launch is then the wall-clock of the timed
            # region / trunk launches in it (everything else charged to the trunk: a lower bound of its efficiency); the
            # kernel alone is exclusive_*.  One lane: the plain average of the event intervals.  
            n_ev = len(trunk_events)
            per_stream_ms = sum(a.elapsed_time(b) for a, b in trunk_events) / n_ev
            launches_per_rank = lanes * (total_sims / world / G) / max(1, args.in_flight)
            ms = (elapsed * 1e3 / launches_per_rank) if lanes > 1 else per_stream_ms
            boards_per_launch = G / float(lanes) * max(1, args.in_flight)
            per_pos = trunk_flops_per_position(cells) if args.evaluator == 'hipnet' else flops_per_position(cells)
            flops = per_pos * boards_per_launch
            achieved = flops / (ms * 1e-3) / 1e12
            peak, pipe_peak, _ = trunk_peak(args)
            pmc_key = line['config']['workload'] + ('+puct' if args.score_mode == 'puct' else '') + \
                ('+k%d' % args.in_flight if args.in_flight > 1 else '') + \
                ('+3launch' if (not deferred_route and args.score_mode != 'puct' and args.in_flight <= 1) else '') + \
                ('+fp8' if args.net_algo == 'split_f16_fp8' else '')
            in_flight = round(per_stream_ms / ms, 2) if lanes > 1 else 1.0
            rf = {'bound': 'mfma', 'kernel': '%s, %d leaves per launch%s' % (
                      evaluator.label, boards_per_launch, (', %.1f launches in flight' % in_flight) if in_flight >= 1.5 else ''),
                  'achieved': round(achieved, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                  'traffic': pmc_traffic('k_trunk', pmc_key),
                  'avg_launch_ms': round(ms, 4), 'avg_launch_ms_per_stream': round(per_stream_ms, 4), 'launches_timed': n_ev,
                  # trunk launches of different lanes share the CUs: how many are in flight on average (a launch's own interval
                  # between its events / the wall-clock per launch); rocprofv3's per-dispatch duration is the former
                  'launches_in_flight': in_flight,
                  'mfma_executed_frac': round(achieved / pipe_peak * executed_flop_ratio(args, cells), 4),
                  'trunk_workgroups': int(min(trunk_wgs if trunk_wgs > 0 else n_cus, boards_per_launch))}
            line['roofline'] = rf
            # ONE dispatch by itself (no other lane on the chip): live, HIP events on the lane's stream around lone launches, and
            # the committed rocprofv3 --kernel-trace --stats average of this very command, which must agree.  A dispatch of B
            # boards holds min(B, CUs) CUs (one 151-KB workgroup per CU): frac_of_chip prices it against the whole chip,
            # frac_of_held_cus against the CUs it holds.  `frac` above is the wall-clock figure with the lanes overlapped.
            held = min(1.0, boards_per_launch / float(n_cus))
            per = {'boards': int(boards_per_launch), 'cus_held': int(round(held * n_cus))}
            if exclusive_ms:
                ex_tf = flops / (exclusive_ms * 1e-3) / 1e12
                per.update({'live_avg_us': round(1e3 * exclusive_ms, 2), 'live_frac_of_chip': round(ex_tf / peak, 4),
                            'live_frac_of_held_cus': round(ex_tf / peak / held, 4)})
            prof = rocprof_dispatch_us('k_trunk_rowsILi15ELb1ELb0ELb0E') if (default_config and args.net_algo == 'split_f16') else None
            if prof and boards_per_launch == 128:
                pr_tf = flops / (prof[0] * 1e-6) / 1e12
                per.update({'rocprof_avg_us': round(prof[0], 2), 'rocprof_calls': prof[1], 'rocprof_stats': prof[2],
                            'rocprof_frac_of_chip': round(pr_tf / peak, 4), 'rocprof_frac_of_held_cus': round(pr_tf / peak / held, 4)})
            rf['per_dispatch'] = per
            if exclusive_ms and boards_per_launch >= n_cus:   # (a launch of fewer boards than CUs cannot fill the chip by itself)
                ex = flops / (exclusive_ms * 1e-3) / 1e12
                rf['exclusive_launch_ms'] = round(exclusive_ms, 4)
                rf['exclusive_frac'] = round(ex / peak, 4)
                if ms > 1.3 * exclusive_ms:
                    rf['eager_samples_host_bound'] = True   # short kernels: the event intervals contain the host's enqueue gaps
            # the two small kernels of a simulation step, bracketed the same way
            fc = [x.elapsed_time(y) for ev in evaluators for x, y in ev.fc_events]
            tr = [x.elapsed_time(y) for ev in evaluators for x, y in ev.tree_events]
            if tr:
                fc_ms, tr_ms = (sum(fc) / len(fc)) if fc else None, sum(tr) / len(tr)
                per_sim = tree_bytes_per_sim(365.5, 208.8, 1.74) if (args.game == 'gomoku' and board == 15) else None
                # (deferred priors: no FC GEMM inside a step -- heads_gemm_ms is None; the tree step finishes the value head itself)
                line['small_kernels'] = {'heads_gemm_ms': round(fc_ms, 4) if fc else None, 'tree_step_ms': round(tr_ms, 4),
                                         'launches_timed': len(tr)}
                if per_sim:
                    # (PUCT scans all children at every level: the same formula with the counts of that rule)
                    gbs = per_sim * boards_per_launch / (tr_ms * 1e-3) / 1e9
                    line['roofline_tree'] = {
                        'bound': 'hbm', 'kernel': '%s, %d games per launch' % ('k_tree_step_def' if deferred_route else 'k_tree_step_raw', boards_per_launch),
                        'achieved': round(gbs, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5),
                        'traffic': pmc_traffic('k_tree_step', pmc_key),
                        'avg_launch_ms': round(tr_ms, 4)}
        elif resident_route:
            # The resident search: ONE launch per search, one workgroup per game running trunk -> value head -> expand / backup ->
            # selection n_playout times.  The dominant work is still the trunk's: its algorithmic flops for the G leaves of a
            # simulation step / the wall-clock per step (tree code, boundaries between moves and host time all charged to it).
            step_ms = elapsed * 1e3 / (total_sims / world / G)
            flops = trunk_flops_per_position(cells) * G
            achieved = flops / (step_ms * 1e-3) / 1e12
            peak, pipe_peak, _ = trunk_peak(args)
            line['roofline'] = {'bound': 'mfma', 'kernel': '%s<RES> (resident search: trunk + value head + tree step of a game in one '
                                'workgroup, one launch per search), %d games' % (evaluator.label, G),
                                'achieved': round(achieved, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                                'traffic': None, 'avg_launch_ms': round(step_ms * args.playouts, 4), 'ms_per_simulation_step': round(step_ms, 5),
                                'launches_per_search': 2, 'trunk_workgroups': G}
            if resident_timing and 'search_ms' in resident_timing:
                # k_delta_res: ONE launch = every simulation of every game of a move.  `achieved` prices the launch the contract's way:
                # the ALGORITHMIC flops of its G x n_playout forward passes (PolicyValueNet.forward on a whole board each: what the
                # reference computes per leaf) / the launch's own duration (HIP events on its stream, lone launches behind the timed
                # regions; the committed rocprofv3 summary of the same command beside it).  The kernel EXECUTES fewer products than that --
                # it recomputes only the windows around the stones a leaf adds to its root (csrc/rz_delta.h) -- and `executed` says
                # how many and at what share of the f16 pipe; tree code, value head and the base of every root are inside the launch.
                rt, st = resident_timing, resident_timing['stats']
                launches = rt['launches']
                leaves = max(1, st['delta'] + st['no_base'])
                alg = trunk_flops_per_position(cells) * G * args.playouts
                ach = alg / (rt['search_ms'] * 1e-3) / 1e12
                mfma_flops = (st['tiles3'] * 16 * 128 * 576 + st['tiles2'] * 16 * 64 * 288) * 2.0 * SPLIT_MFMAS_PER_PRODUCT / launches
                prof = rocprof_dispatch_us('k_delta_res')
                pmc_key = line['config']['workload']
                rf = {'bound': 'mfma',
                      'kernel': 'k_delta_res (resident search with the receptive-field trunk: trunk windows + value head + tree step of a game in one '
                                'workgroup, two games per CU), %d games x %d simulations per launch' % (G, args.playouts),
                      'achieved': round(ach, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                      'traffic': pmc_traffic('k_delta_res', pmc_key),
                      'avg_launch_ms': round(rt['search_ms'], 4), 'launches_timed': launches, 'bases_launch_ms': round(rt['bases_ms'], 4),
                      'algorithmic_flops_per_launch': alg,
                      'note': 'achieved / frac price the ALGORITHMIC flops (a whole-board forward per leaf, the contract\'s figure) and can pass 1: '
                              'the kernel executes only `executed.share_of_the_algorithmic_products` of them -- `executed` is the pipe\'s own rate',
                      'executed': {'f16_mfma_tflops': round(mfma_flops / (rt['search_ms'] * 1e-3) / 1e12, 2),
                                   'frac_of_f16_mfma_peak': round(mfma_flops / (rt['search_ms'] * 1e-3) / 1e12 / pipe_peak, 4),
                                   'share_of_the_algorithmic_products': round(mfma_flops / SPLIT_MFMAS_PER_PRODUCT / alg, 4),
                                   'conv3_tiles_per_leaf': round(st['tiles3'] / leaves, 3), 'conv2_tiles_per_leaf': round(st['tiles2'] / leaves, 3),
                                   'changed_cells_per_leaf': round(st['cells'] / leaves, 3),
                                   'leaves_against_a_base': st['delta'], 'leaves_without_a_base': st['no_base']},
                      'wall_clock': {'ms_per_simulation_step': round(step_ms, 5), 'achieved': round(achieved, 3), 'frac': round(achieved / peak, 4),
                                     'note': 'the timed region / simulation steps in it: move steps, base builds, policy GEMM and host time charged to the search'},
                      'launches_per_search': 1, 'trunk_workgroups': G, 'workgroups_per_cu': 2}
                if prof:
                    rf['rocprof_avg_ms'] = round(prof[0] / 1e3, 4)
                    rf['rocprof_calls'], rf['rocprof_stats'] = prof[1], prof[2]
                    rf['rocprof_frac'] = round(alg / (prof[0] * 1e-6) / 1e12 / peak, 4)
                ghz = st.get('resident_sclk_ghz')
                if ghz:
                    rf['sclk_in_loop_ghz'] = round(ghz, 3)   # (shader cycles / constant-clock ticks of workgroup 0 of the last timed launch)
                    rf['value_at_2p0ghz'] = round(value * 2.0 / ghz, 1)
                    rf['us_per_simulation_of_a_game'] = round(rt['search_ms'] * 1e3 / args.playouts, 3)
                    rf['trunk_workgroup_us'] = rf['us_per_simulation_of_a_game']   # (a workgroup's trunk windows + value head + tree step of ONE leaf: the resident kernel has no trunk launch of its own)
                rf['board_power_w'] = power.mean()
                line['roofline'] = rf
                if args.game == 'gomoku' and board == 15:
                    # the tree code has no launch of its own any more (it is the serial part of k_delta_res' workgroups): its ALGORITHMIC
                    # bytes (7.86 KB per simulation in the reference's dense formulation) over the same launch -- what
                    # north_star calls the tree traversal's fraction of the HBM roofline; the launch's counter traffic is `roofline.traffic`
                    per_sim = tree_bytes_per_sim(365.5, 208.8, 1.74)
                    gbs = per_sim * G * args.playouts / (rt['search_ms'] * 1e-3) / 1e9
                    line['roofline_tree'] = {'bound': 'hbm', 'kernel': 'the tree code inside k_delta_res (expand_backup_body + select_body of rz_tree.h, one wave per game)',
                                             'achieved': round(gbs, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 5),
                                             'traffic': None, 'avg_launch_ms': round(rt['search_ms'], 4),
                                             'note': 'latency-bound by construction: ~20 k of a simulation\'s 58 k cycles are one wave walking one tree (profiles/r06/delta_resident_phases.txt)'}
            elif resident_timing:
                line['roofline']['resident_timing_error'] = resident_timing.get('error')
        else:
            per_sim = tree_bytes_per_sim(365.5, 208.8, 1.74) if board == 15 else None
            if per_sim:
                achieved = value / world * per_sim / 1e9
                line['roofline'] = {'bound': 'hbm', 'kernel': 'k_select + k_expand_backup (tree only)',
                                    'achieved': round(achieved, 3), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                    'frac': round(achieved / PEAK_HBM_GBS, 6), 'traffic': None}
        line['fill_%d' % FILL_GAMES_PER_GPU] = fill
        line['configs'] = config_legs
        line['cpu_baseline'] = cpu_baseline
    for eng in engines:
        eng.close()
    if rank == 0:
        # The schedule without a profiler in the way (rlzero_amd/trace.py): the same layout played by a CHILD process (before this
        # one touched the GPU) with the device-side launch trace attached -- every trunk / tree-step workgroup leaves its start, end
        # and CU.  launches_in_flight, the CUs' time under trunk workgroups and the lanes' step cycle come from those records, not
        # from a ratio of averaged event intervals (rocprofv3 serialises the lanes' queues: profiles/r03/trunk_overlap_default.json).
        if timeline is not None and deferred_route and not resident_route and lanes > 1 and 'roofline' in line:
            line['lane_timeline'] = timeline
            if 'launches_in_flight' in timeline:
                line['roofline']['launches_in_flight_from_event_averages'] = line['roofline']['launches_in_flight']
                line['roofline']['launches_in_flight'] = timeline['launches_in_flight']
                line['roofline']['cu_time_in_trunk'] = timeline['cu_time_in_trunk']
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
