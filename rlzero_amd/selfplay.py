"""Many self-play games in lock-step on one GPU, sharded over GPUs by game id.

Batched counterpart of ``GameControl.start_self_play`` (rlzero/games/gomoku/game.py:96-134)
+ ``AlphaZeroPlayer.get_action`` (rlzero/mcts/alphazero_mcts.py:136-165): every game does
exactly what the reference does for one game -- n_playout simulations from the current
root, pi = softmax(log(N + 1e-10) / T) over the legal moves, a move drawn from pi, tree
reuse, z from the final winner -- but the searches of all games advance together so each
simulation step is ONE batch for the kernels and the network.

Randomness: the reference draws the move with ``numpy.random.choice(acts, p=probs)`` from
the global stream (alphazero_mcts.py:148), i.e. ``acts[searchsorted(cdf, u, 'right')]`` for
the next uniform ``u``.  Here ``u`` comes from a counter-based generator keyed by
(seed, game id, ply), so a game's trajectory does not depend on which GPU plays it or on
how many games share the batch.

Multi-GPU: game ``g`` belongs to rank ``g % world_size``; there is no collective inside
the search.  ``gather_trajectories`` is the single exchange: fixed-stride records to
rank 0 (RCCL when the process group backend is nccl, gloo in the CPU tests).
"""
import numpy as np

from ._hip import HipError

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def move_uniform(seed, game_id, ply):
    """Uniform in [0,1) for (seed, game, ply): 53 high bits of a splitmix64 chain."""
    with np.errstate(over='ignore'):
        x = _splitmix64(np.uint64(seed))
        x = _splitmix64(x ^ np.asarray(game_id, dtype=np.uint64))
        x = _splitmix64(x ^ np.asarray(ply, dtype=np.uint64))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def visits_to_pi(counts, temperature):
    """alphazero_mcts.py:10-14,91-92 on the visit counts of the legal moves."""
    x = 1.0 / temperature * np.log(np.asarray(counts) + 1e-10)
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


def draw_move(acts, probs, u):
    """numpy's legacy choice(acts, p=probs) for the uniform ``u``."""
    cdf = np.cumsum(probs)
    cdf /= cdf[-1]
    return int(acts[int(cdf.searchsorted(u, side='right'))])


def batch_pi_and_moves(visits, legal, temperature, uniforms):
    """visits_to_pi + draw_move for many games at once: visits int [R, A], legal bool [R, A], uniforms [R]
    -> (pi float64 [R, A] with zeros at illegal actions, chosen action int [R]).

    Bit-identical to the per-game expressions above (which are the reference's, alphazero_mcts.py:10-14,
    91-92,148): log / exp / divide are elementwise; the maximum is exact; the normalising sum is taken over
    each game's COMPRESSED legal entries, grouped by their count, because numpy's pairwise summation depends on
    the array length; cumsum is sequential and adding the zeros of illegal actions changes nothing, so the
    first index with cdf > u on the full row is searchsorted(u, 'right') on the compressed one."""
    visits = np.asarray(visits)
    legal = np.asarray(legal, dtype=bool)
    R, A = visits.shape
    x = 1.0 / temperature * np.log(visits + 1e-10)
    mx = np.where(legal, x, -np.inf).max(axis=1)
    e = np.exp(np.where(legal, x - mx[:, None], -np.inf))  # exp(-inf) = 0 at illegal actions
    k = legal.sum(axis=1)
    flat = e[legal]                      # every game's legal entries, ascending, one game after the other
    start = np.cumsum(k) - k
    sums = np.ones(R)
    for kk in np.unique(k):
        if kk > 0:
            rows = np.nonzero(k == kk)[0]
            sums[rows] = flat[start[rows][:, None] + np.arange(kk)].sum(axis=1)   # (a contiguous [rows, kk] array: numpy's pairwise sum of kk)
    pi = e / sums[:, None]
    cdf = np.cumsum(pi, axis=1)
    cdf /= cdf[:, -1:]
    moves = (cdf > np.asarray(uniforms, dtype=np.float64)[:, None]).argmax(axis=1)
    return pi, moves.astype(np.int64)


def shard_game_ids(n_games_total, rank, world_size):
    return list(range(rank, n_games_total, world_size))


class Trajectory(object):
    """One finished game: what start_self_play returns, in compact form."""

    def __init__(self, game_id, board_size, n_in_row, moves, pis, winner, game='gomoku'):
        self.game_id = int(game_id)
        self.game = game
        self.board_size, self.n_in_row = board_size, int(n_in_row)
        self.moves = [int(m) for m in moves]
        pis = np.asarray(pis)
        if pis.dtype != np.float32:   # (float32 rows -- what came over the wire for the learner -- stay as they are: no 2x copy of a round's pi)
            pis = pis.astype(np.float64, copy=False)
        self.pis = pis.reshape(len(self.moves), -1) if len(self.moves) else pis.reshape(0, 0)
        self.winner = int(winner)

    def _env(self):
        if self.game == 'connect4':
            from .games.connect4.connect4_env import Connect4Env
            return Connect4Env(self.board_size[0], self.board_size[1], self.n_in_row)
        from .games.gomoku.gomoku_env import GomokuEnv
        env = GomokuEnv(self.board_size, self.n_in_row)
        env.reset()
        return env

    def z(self):
        """+1 for the plies of the winner, -1 for the loser's, 0 on a tie (game.py:121-126)."""
        movers = np.arange(len(self.moves)) % 2  # player 0 moves first
        if self.winner == -1:
            return np.zeros(len(self.moves))
        return np.where(movers == self.winner, 1.0, -1.0)

    def states(self):
        """Observation planes before every move (GomokuEnv.current_state, gomoku_env.py:95-114): own stones, the opponent's, the
        last move, ones iff an even number of stones.  Gomoku: formed from the move list at once (ply q's stone is on the board of
        ply p > q, in plane 0 when q and p have the same parity -- two products of 0 / 1 matrices); Connect4 (stones drop: the cell
        depends on the column's height) replays the moves through the env."""
        if self.game == 'connect4':
            env = self._env()
            out = []
            for m in self.moves:
                out.append(env.current_state())
                env.step(m)
            return out
        P, B = len(self.moves), self.board_size
        if P == 0:
            return []
        moves = np.asarray(self.moves, dtype=np.int64)
        onehot = np.zeros((P, B * B))
        onehot[np.arange(P), moves] = 1.0
        before = np.tri(P, P, -1)                                    # [p, q] = 1 where ply q was played before ply p
        parity = np.arange(P) % 2
        same = (parity[:, None] == parity[None, :]).astype(np.float64)
        planes = np.zeros((P, 4, B * B))
        planes[:, 0] = (before * same) @ onehot
        planes[:, 1] = (before * (1.0 - same)) @ onehot
        planes[1:, 2] = onehot[:-1]
        planes[0::2, 3] = 1.0
        return list(planes.reshape(P, 4, B, B))

    def as_reference_tuple(self):
        """(winner, [(state, mcts_prob, z), ...]) -- start_self_play's return value."""
        return self.winner, list(zip(self.states(), list(self.pis), self.z()))




def plan_lanes(n_games, n_cus=256, hw_queues=None, deferred=False, cells=None, in_flight=1, resident_per_cu=1):
    """-> (lanes, trunk_workgroups, heads_algo) for ``n_games`` leaves per simulation step on a GPU with ``n_cus`` CUs and
    ``hw_queues`` hardware queues for its streams (default: what rlzero_amd claimed on import, rlzero_amd.HW_QUEUES).

    ``deferred``: the batch runs the deferred-priors route (HipNetEvaluator.deferred_ok: UCT_REF, one simulation in flight, boards
    of 11 .. 16 rows).  A lane's step is then trunk -> tree step (23 + 10 us at 15x15) and the table was measured again
    (profiles/r04/lane_sweep.txt, M simulations / s): up to one round of boards the resident search, one workgroup per game and one
    launch per search, on one lane up to half a round and on two beyond (256 games: 9.2 against 7.8 on two lanes of the two-launch
    step, profiles/r04/ab_resident.txt); up to
    1.75 rounds THREE (320 / 384 games: 9.2 / 10.3 against 8.7 / 9.8 with
    two); 448 games TWO (10.3 against 10.1); 2 .. 2.75 rounds FOUR on 8 hardware queues (512 / 640 games: 10.6 / 10.6 against 10.5
    / 10.1 with two -- with fewer queues two lanes, a percent behind); beyond, TWO (768 .. 1536 games: 10.9 .. 11.0).

    ``resident_per_cu`` = 2 (with ``deferred``: the receptive-field trunk, HipNetEvaluator.resident_delta_ok -- boards of 11 .. 16 rows
    and columns): the resident search holds TWO games per CU (k_delta_res: 82 KB of LDS), so up to 2 x CUs games -- the 512 per GPU of
    BASELINE.json configs[3] -- are ONE launch per search on ONE lane: one game's serial tree walk runs under the other game's matrix
    work on the same CU, and no hardware-queue layout is involved (round 6, same box: 512 games 18.95 M on one lane, 17.0 M on two,
    13.7 M on the four lanes of the two-launch step; profiles/r06/NOTES.md).  From 3 x CUs games on the same single lane again, the
    grid running in rounds of 2 x CUs workgroups (1024 .. 4096 games: 18.4 .. 18.8 M; profiles/r06/ab_rounds.txt).

    ``cells``: positions of the board, when known.  Boards of at most 42 cells on the split-f16 trunk (Connect4, 6x6: the
    two-launch step with a 12-us trunk and a 10-us tree step) run TWO lanes above one round of boards (Connect4, M simulations / s on
    2 / 3 / 4 lanes: 384 games 14.2 / 13.8 / 13.1, 512 games 17.7 / 15.7 / 16.7, 768 games 20.2 / 20.4 / 20.0, 1024 games
    22.9 / 21.9 / 21.1 -- profiles/r04/small_boards_lanes.txt; 9x9 keeps the table below: 512 games 14.8 / 14.5 / 15.4).

    ``in_flight`` = K > 1 (the opt-in virtual-loss mode; ``n_games`` = games x K leaves) on a board of at most 100 cells: the
    three-launch step of short kernels wants MORE lanes than the 15x15 table below gives it (9x9, K = 16, M simulations / s on
    1 / 2 / 3 / 4 lanes: 32 games 4.5 / 4.7 / 5.1 / 4.3, 64 games 6.8 / 7.4 / 7.9 / 8.1, 128 games 8.6 / 10.0 / 10.7 / 11.2; K = 8,
    64 games 6.0 / 6.5 / 6.9 / 7.0): three lanes from two rounds of leaves, four from four rounds.

    The three-launch step (every other batch), measured on
    MI355X at 15x15 (profiles/r03/lane_sweeps.txt; a trunk workgroup takes a board in ~23 us, three in ~65 us):

    * up to one round of boards (n_games <= CUs): ONE lane -- the step is a chain of three latency-bound launches, and splitting it
      shortens none of them (128 / 192 / 256 games: 3.4 / 4.9 / 6.3 M with one lane, 3-8 % less with two to four);
    * more: lanes with UN-capped trunks and the LDS-free 'parts' FC GEMM, whose single-wave workgroups -- like the tree step's -- fit
      on a CU beside a resident trunk workgroup (the trunk holds 151 of 160 KB LDS and at most 400 of 512 registers): the small kernels
      of a lane run UNDER the other lanes' trunks on the same CUs.  With TWO lanes a lane's tree step + FC GEMM (15 us + two kernel
      boundaries) must end before the other lane's trunk does (23 us at one board per workgroup), or the CUs wait: on most boxes they
      do (512 games: 8.0-8.9 M, on the best box 10.3 M).  FOUR lanes of a quarter of the games keep two lanes' trunk workgroups queued
      on the CUs at all times and the chip never waits for a lane's chain: 9.8-10.3 M on every box -- from ~1.75 to ~2.75 rounds of
      boards (448 .. 704 games), and only with 8 hardware queues (with HIP's default of 4 the fourth lane shares a queue and the lanes
      serialise: 6.1 M; five lanes and more collapse the same way even with 16);
    * THREE lanes between one and 1.75 rounds (320 / 384 games: +4 % / +1 % over two) and wherever four would need more queues;
    * beyond ~2.75 rounds TWO lanes: a trunk launch then runs two or three boards per workgroup (44 / 65 us) and hides a lane's chain
      by itself (768 / 1024 / 1536 games: 10.1 / 11.3 / 11.4 M with two lanes, 9.5 / 9.1 / 10.2 with four).
    ``trunk_workgroups`` is 0 = one per CU (capped trunks, 4 CUs per XCD left to the small kernels, lost 15-19 % against un-capped
    ones once the 'parts' GEMM existed)."""
    if hw_queues is None:
        from . import HW_QUEUES as hw_queues
    if cells is not None and cells <= 49 and n_cus < n_games <= 2 * n_cus and in_flight <= 1 and resident_per_cu >= 2:
        # boards of the compact LDS grid (up to 7 columns; HipNet.compact_resident): the resident search holds two games per CU --
        # one launch per search on one lane up to 2 x CUs games (profiles/r06/ab_compact_resident.txt, M simulations / s against
        # the two lanes below: Connect4 384 games 19.8 / 17.1, 512 games 24.8 / 21.3; 6x6 512 games 26.5 / 22.5).  Beyond, a
        # partial second round costs more than the lanes' overlap (768 games 21.2 / 24.7; 1024 25.7 / 25.0; 2048 26.5 / 27.6)
        return 1, 0, 'auto'
    if cells is not None and cells <= 42 and n_games > n_cus and in_flight <= 1:
        return 2, 0, 'parts'
    if cells is not None and cells <= 100 and in_flight > 1 and n_games >= 2 * n_cus:
        return (4 if (n_games >= 4 * n_cus and hw_queues >= 8) else 3), 0, 'parts'
    if deferred:
        if resident_per_cu >= 2 and (n_games <= 2 * n_cus or n_games >= 3 * n_cus):
            # two games per CU: the resident search with the receptive-field trunk, one lane -- and from 1.5 rounds of two per CU on
            # again: the launch's workgroups depend on nothing outside their game, so a bigger grid runs in ROUNDS, a CU's free half
            # going to the next game as a search ends (profiles/r06/ab_rounds.txt, M simulations / s, one resident lane against the
            # lanes below: 576 games 12.4 / 14.0, 640 13.6 / 14.5, 768 14.6 / 12.6, 1024 18.8 / 11.9, 1536 18.6 / 14.0, 2048 18.4 / 16.1,
            # 4096 18.4 / 16.9 -- between one and 1.5 rounds the second round is too empty and four lanes of the two-launch step win)
            return 1, 0, 'auto'
        if n_games <= n_cus:   # one game per CU at most: the resident search (one launch per search, a workgroup per game) --
            # on TWO lanes from half a round of boards on, so that a lane's host step runs under the other lane's search
            # (256 games: 9.17 against 8.57 M; 128 games 4.83 against 4.74; profiles/r04/ab_resident.txt)
            return (2 if 2 * n_games > n_cus else 1), 0, 'auto'
        if 4 * n_games < 7 * n_cus:
            return 3, 0, 'parts'
        if n_games < 2 * n_cus or 4 * n_games > 11 * n_cus:
            return 2, 0, 'parts'
        return (4 if hw_queues >= 8 else 2), 0, 'parts'
    if n_games <= n_cus:
        return 1, 0, 'auto'
    if 4 * n_games < 7 * n_cus:
        return 3, 0, 'parts'
    if 4 * n_games <= 11 * n_cus:
        if hw_queues < 8:
            import warnings
            from . import HW_QUEUES_TOO_LATE
            warnings.warn('plan_lanes: %d games want four lanes, but this process has %d hardware queues%s: three lanes (about 8 %% '
                          'slower at 512 games); see rlzero_amd.configure' % (
                              n_games, hw_queues, ' (rlzero_amd was imported after the HIP runtime had started)'
                              if HW_QUEUES_TOO_LATE else ''), RuntimeWarning, stacklevel=2)
            return 3, 0, 'parts'
        return 4, 0, 'parts'
    return 2, 0, 'parts'


# ------------------------------------------------------------------------- lanes by measurement
# plan_lanes() is a table measured on ONE chip (MI355X, 256 CUs, 8 hardware queues) for the reference's network: its thresholds are
# in units of that chip's CUs and its neighbouring choices differ by up to 15 %.  Anywhere else -- a partition of the GPU with
# fewer CUs, another chip -- the table's pick is only the starting point: the two or three neighbouring layouts are built, timed
# for two moves each and the fastest is kept, once per (device, CUs, board, batch, search) in this process.
TABLE_CUS = 256
_LANE_CACHE = {}


def lane_candidates(table_pick, hw_queues, n_games):
    """The layouts worth timing around the table's pick: one lane fewer, the pick, one more (four lanes need 8 hardware queues)."""
    most = min(4 if hw_queues >= 8 else 3, max(1, n_games))
    return sorted({min(most, max(1, table_pick - 1)), min(most, max(1, table_pick)), min(most, table_pick + 1)})


def time_moves(sp, moves=2):
    """Simulations / second of ``sp`` over ``moves`` moves of fresh games (one more move first, un-timed), the move step on the device."""
    import time
    t = sp.torch
    sp.device_attach(queue_capacity=4 * sp.n_slots)
    sp.device_queue(range(4 * sp.n_slots))
    sp.play_move_device()
    t.cuda.synchronize()
    sp.device_drain()
    s0, t0 = sp.sims_done, time.perf_counter()
    for _ in range(moves):
        sp.play_move_device()
    t.cuda.synchronize()
    sp.device_drain()
    return (sp.sims_done - s0) / (time.perf_counter() - t0)


def choose_lanes_by_measurement(key, table_pick, hw_queues, n_games, build, timer=time_moves, cache=None):
    """-> (lanes, {lanes: simulations / s}) for ``key``: every candidate layout is built by ``build(lanes)`` (-> a BatchedSelfPlay),
    timed by ``timer(sp)`` and closed; the fastest wins, ties go to fewer lanes.  Cached by ``key``."""
    cache = _LANE_CACHE if cache is None else cache
    if key in cache:
        return cache[key]
    rates = {}
    for lanes in lane_candidates(table_pick, hw_queues, n_games):
        sp = None
        try:   # (a candidate that cannot be built or timed -- e.g. a mode the timer's move step does not serve -- is no candidate)
            sp = build(lanes)
            rates[lanes] = float(timer(sp))
        except HipError:
            pass
        finally:
            for lane in getattr(sp, 'lanes', ()) if sp is not None else ():
                lane.eng.close()
    if not rates:   # nothing could be measured: the table's pick, not cached as a measurement
        return table_pick, None
    best = max(sorted(rates), key=lambda c: (rates[c], -c))
    cache[key] = (best, rates)
    return cache[key]


def fc_in_trunk_pays(rows, cols, n_actions):
    """True when the trunk's workgroups should run the first FC layers on their own boards (HipNet.set_heads_algo('in_trunk'))
    instead of a GEMM launch of its own: boards of up to 10 rows whose FC weights (hi + lo f16) are at most 40 KB -- every workgroup
    streams them for its one board (6x6: 39 KB, TicTacToe +6 %; Connect4: 26 KB, 512 games on two lanes +3 %; 9x9: 146 KB, -5 to
    -11 %: profiles/r03/in_trunk_fc.txt)."""
    cells = rows * cols
    return rows <= 10 and (n_actions * 4 * cells + 64 * 2 * cells) * 4 <= 40 * 1024


class _Lane(object):
    """One engine + its evaluator + the HIP stream its kernels are enqueued on."""

    def __init__(self, engine, evaluator, stream, offset):
        self.eng, self.evaluator, self.stream, self.offset = engine, evaluator, stream, offset
        self.slots = slice(offset, offset + engine.n_games)


class BatchedSelfPlay(object):
    """Plays games on one GPU; slots are refilled as games end.

    ``engine`` / ``evaluator`` may be lists of equal length: each (engine, evaluator) pair is a
    *lane* with its own HIP stream, and the simulation steps of the lanes are enqueued
    alternately.  While the network kernel of one lane owns the CUs' LDS, the latency-bound tree
    kernels of the other lane run beside it (they use no LDS), so the halves ping-pong and the
    tree work is hidden under the dense contraction.  Results do not depend on the lane split:
    games are independent and their uniforms are keyed by game id."""

    def __init__(self, engine, evaluator, temperature=1.0, seed=0, use_graph=False, sims_per_graph=8,
                 eager_every=0):
        engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
        evaluators = list(evaluator) if isinstance(evaluator, (list, tuple)) else [evaluator]
        assert len(engines) == len(evaluators) >= 1
        torch = engines[0].torch
        self.torch = torch
        self.lanes, offset = [], 0
        for i, (eng, ev) in enumerate(zip(engines, evaluators)):
            stream = torch.cuda.current_stream(eng.device) if len(engines) == 1 else \
                torch.cuda.Stream(device=eng.device)
            # the engine's buffers were initialised on the current stream: order the lane's stream behind it
            stream.wait_stream(torch.cuda.current_stream(eng.device))
            self.lanes.append(_Lane(eng, ev, stream, offset))
            offset += eng.n_games
        n_cus = torch.cuda.get_device_properties(engines[0].device).multi_processor_count

        def per_cu(ev, eng):   # resident workgroups a CU holds: two of the receptive-field kernel (k_delta_res) and of the compact grid's
            inner = getattr(ev, 'inner', ev)
            fn = getattr(inner, 'resident_per_cu', None)
            return fn(eng) if fn is not None else 1
        if len(engines) > 1 and sum(e.n_games for e in engines) > n_cus * min(per_cu(ev, e) for ev, e in zip(evaluators, engines)):
            # lanes that share CUs: the resident search (a workgroup keeps its CU for a whole search) is for games that have a CU
            # each -- these lanes run the two-launch step, whose trunk workgroups make way for the other lanes every step
            for ev in evaluators:
                inner = getattr(ev, 'inner', ev)   # (bench.py wraps its evaluators)
                if hasattr(inner, 'resident_search'):
                    inner.resident_search = False
        self.eng = engines[0]  # geometry (board size, n_playout) is common to all lanes
        self.evaluator = evaluators[0]
        self.n_slots = offset
        self.temperature = float(temperature)
        self.seed = int(seed)
        self.use_graph = use_graph
        self.sims_per_graph = sims_per_graph
        self.eager_every = int(eager_every)  # with graphs: every k-th chunk runs eagerly (timing samples)
        G = self.n_slots
        self.slot_game = np.full(G, -1, dtype=np.int64)
        self.slot_ply = np.zeros(G, dtype=np.int64)
        self.cell_taken = np.zeros((G, self.eng.n_cells), dtype=bool)  # host mirror of the root boards
        self.slot_moves = [[] for _ in range(G)]
        self.slot_pis = [[] for _ in range(G)]
        self.sims_done = 0
        self.moves_done = 0

    @classmethod
    def for_network(cls, net_module, board, n_in_row, n_games, n_playout, c_puct=5.0, device='cuda:0',
                    game='gomoku', net_shape=None, lanes=None, trunk_workgroups=None, temperature=1.0, seed=0,
                    use_graph=True, sims_per_graph=16, eager_every=0, add_noise=True, sims_in_flight=1, before_warm=None,
                    deferred_priors=None, resident_search=None, net_algo=None, delta_trunk=None, **engine_kw):
        """Self-play of ``n_games`` games in flight with the hand-written evaluator of ``net_module`` (a
        PolicyValueNet): builds the lanes (engine + HipNetEvaluator each) as plan_lanes() recommends, unless
        ``lanes`` / ``trunk_workgroups`` are given (more than four lanes take turns on the GPU's four compute pipes, and four need
        GPU_MAX_HW_QUEUES >= 8: profiles/r03/lane_sweeps.txt).  ``add_noise``: Dirichlet noise on the priors of every expanded
        node, what ``AlphaZeroPlayer(is_selfplay=True)`` does (alphazero_mcts.py:124-129, node.py:63-69).
        ``lanes``: None = the table (plan_lanes) on the chip it was measured on, the measuring fallback elsewhere (another CU count:
        the table's pick and its neighbours are built and timed for two moves each, choose_lanes_by_measurement); 'measure' forces the
        measurement, 'table' the table; an int is taken as given.
        ``sims_in_flight`` = K > 1: the opt-in virtual-loss mode (MCTSEngine), for batches too small to fill the GPU
        with one leaf per game; the evaluator batch of a lane is then its games x K.  ``deferred_priors``: None = the deferred-priors
        route wherever it exists (HipNetEvaluator.deferred_ok), False = the three-launch step everywhere; ``resident_search`` likewise for
        the one-launch-per-search kernel of batches that give every game a CU (False = the two-launch step).  ``before_warm(sp)``: called
        before the hipGraphs are captured (rlzero_amd.trace attaches its buffer there).  ``net_algo``: HipNet.set_algo for every lane's
        evaluator -- None keeps the default ('split_f16', the f32-accurate trunk); 'split_f16_fp8' is the OPT-IN arithmetic narrower than
        the reference's f32 (boards of 11 .. 16 rows and columns).  ``delta_trunk``: False = the full-board trunk on every leaf (the
        checker of the receptive-field evaluation, HipNetEvaluator.delta_trunk; with it goes the resident search's second game per CU)."""
        import torch
        from .engine import HipNetEvaluator, MCTSEngine
        dev = torch.device(device)
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        K = max(1, int(sims_in_flight))
        shape0 = net_shape if net_shape is not None else board
        rows0, cols0 = (shape0[0], shape0[1]) if isinstance(shape0, (tuple, list)) else (shape0, shape0)
        deferred = (deferred_priors is not False and K == 1 and engine_kw.get('score_mode', 'uct_ref') in ('uct_ref', 0)
                    and game == 'gomoku' and 11 <= rows0 <= 16 and 11 <= cols0 <= 16
                    and net_algo in (None, 'split_f16', 'split_f16_tiles', 'split_f16_fp8'))
        small_trunk = (K == 1 and deferred_priors is not False and engine_kw.get('score_mode', 'uct_ref') in ('uct_ref', 0)
                       and net_algo in (None, 'split_f16', 'split_f16_tiles'))   # (the two-launch step on a small board)
        import os
        delta_res = (deferred and resident_search is not False and delta_trunk is not False and net_algo in (None, 'split_f16')
                     and os.environ.get('RZ_NET_DELTA', '1') != '0' and os.environ.get('RZ_NET_DELTA_RESIDENT', '1') != '0')
        from .engine import compact_grid_board
        compact_res = (small_trunk and not deferred and resident_search is not False and net_algo in (None, 'split_f16')
                       and max(rows0, cols0) <= 10 and compact_grid_board(rows0, cols0))
        auto_lanes, auto_wgs, heads_algo = plan_lanes(n_games * K, n_cus, deferred=deferred,
                                                      cells=rows0 * cols0 if (small_trunk or K > 1) else None, in_flight=K,
                                                      resident_per_cu=2 if (delta_res or compact_res) else 1)
        measured = None
        if lanes == 'table':
            lanes = None
        elif K > 1:
            # K simulations in flight: the timer's move step (rz_play_attach) serves one simulation in flight per tree only -- the table
            lanes = None if lanes == 'measure' else lanes
        elif lanes == 'measure' or (lanes is None and n_cus != TABLE_CUS and n_games * K > n_cus):
            # off the table's chip (or asked for): the table's pick and its neighbours, timed for two moves each
            from . import HW_QUEUES
            key = (torch.cuda.get_device_name(dev), n_cus, rows0, cols0, game, n_games, n_playout, K, bool(deferred), str(net_algo))

            def build(n_lanes):
                return cls.for_network(net_module, board, n_in_row, n_games, n_playout, c_puct=c_puct, device=device, game=game,
                                       net_shape=net_shape, lanes=n_lanes, trunk_workgroups=trunk_workgroups, temperature=temperature,
                                       seed=seed, use_graph=use_graph, sims_per_graph=sims_per_graph, add_noise=add_noise,
                                       sims_in_flight=sims_in_flight, deferred_priors=deferred_priors, resident_search=resident_search,
                                       net_algo=net_algo, delta_trunk=delta_trunk, **engine_kw)
            lanes, measured = choose_lanes_by_measurement(key, auto_lanes, HW_QUEUES, n_games, build)
        if lanes is None:
            lanes, wgs = auto_lanes, auto_wgs
        else:
            wgs, heads_algo = (0, 'parts') if lanes > 1 else (0, 'auto')
        if trunk_workgroups is not None:
            wgs = trunk_workgroups
        shape = net_shape if net_shape is not None else board
        rows, cols = (shape[0], shape[1]) if isinstance(shape, (tuple, list)) else (shape, shape)
        acts = shape[2] if isinstance(shape, (tuple, list)) and len(shape) > 2 else rows * cols
        if heads_algo == 'parts' and int(wgs) == 0 and fc_in_trunk_pays(rows, cols, acts):
            heads_algo = 'in_trunk'   # small FC layers: no GEMM launch at all (each lane's chain loses a kernel and a boundary)
        lanes = max(1, min(int(lanes), n_games))
        per_lane = [n_games // lanes + (1 if i < n_games % lanes else 0) for i in range(lanes)]
        engines, evaluators = [], []
        for g_lane in per_lane:
            engines.append(MCTSEngine(board, n_in_row, n_games=g_lane, n_playout=n_playout, c_puct=c_puct,
                                      device=str(device), game=game, add_noise=add_noise, sims_in_flight=K,
                                      noise_seed=(int(seed) * 7919 + len(engines)) & 0x7fffffff, **engine_kw))
            ev = HipNetEvaluator(net_module, net_shape if net_shape is not None else board, str(device),
                                 max_boards=g_lane * K)
            if net_algo is not None:
                ev.hip.set_algo(net_algo)
            ev.hip.set_max_workgroups(max(0, int(wgs)))
            ev.hip.set_heads_algo(heads_algo)
            if deferred_priors is not None:
                ev.deferred_priors = bool(deferred_priors)
            if resident_search is not None:
                ev.resident_search = bool(resident_search)
            if delta_trunk is not None:
                ev.delta_trunk = bool(delta_trunk)
            evaluators.append(ev)
        sp = cls(engines if lanes > 1 else engines[0], evaluators if lanes > 1 else evaluators[0],
                 temperature=temperature, seed=seed, use_graph=use_graph, sims_per_graph=sims_per_graph,
                 eager_every=eager_every)
        sp.trunk_workgroups = int(wgs)
        sp.lanes_measured = measured   # {lanes: simulations / s} when the layout was chosen by measurement, else None
        if before_warm is not None:
            before_warm(sp)
        sp.warm_graphs()
        return sp

    def warm_graphs(self):
        """Capture the simulation-chunk hipGraph of every lane (before any game is started: the capture
        runs the chunk once).  Weight updates keep the graphs valid: rz_net_load reuses its device buffers."""
        if not self.use_graph:
            return
        per = self.eng.graph_chunk(self.sims_per_graph)
        for lane in self.lanes:
            with self._on(lane):
                lane.eng.reset_games()
                lane.eng.warm_graph(lane.evaluator, per)
        self.torch.cuda.synchronize()

    def refresh_weights(self):
        """Re-upload the network weights of every lane if the torch module changed (after a learner step)."""
        for lane in self.lanes:
            refresh = getattr(lane.evaluator, 'refresh_if_changed', None)
            if refresh is not None:
                refresh(content=True)

    def _on(self, lane):
        return self.torch.cuda.stream(lane.stream)

    # -- slot management -----------------------------------------------------------
    def _start(self, slots, game_ids):
        mask = np.zeros(self.n_slots, dtype=np.uint8)
        for s, g in zip(slots, game_ids):
            mask[s] = 1
            self.slot_game[s] = g
            self.slot_ply[s] = 0
            self.cell_taken[s] = False
            self.slot_moves[s] = []
            self.slot_pis[s] = []
        # a game's Dirichlet noise (read by the PUCT rule only), like its move draws, is keyed by (seed, game id): the trajectory
        # does not depend on the slot, lane or GPU the game is played on
        with np.errstate(over='ignore'):
            keys = _splitmix64(_splitmix64(np.uint64(self.seed) ^ np.uint64(0x6E6F697365000000)) ^ self.slot_game.astype(np.uint64))
        for lane in self.lanes:
            if mask[lane.slots].any():
                if getattr(lane, 'primed', False):
                    # a pipelined run that stopped early (max_moves) left the next move's simulations enqueued: they end
                    # on the old roots, and the trees reset here must not be taken for searched ones
                    lane.stream.synchronize()
                    lane.primed = False
                with self._on(lane):
                    lane.eng.reset_games(mask=mask[lane.slots])
                    lane.eng.set_noise_keys(keys[lane.slots], mask=mask[lane.slots])

    def _set_active(self):
        active = (self.slot_game >= 0).astype(np.uint8)
        for lane in self.lanes:
            with self._on(lane):
                lane.eng.set_active(active[lane.slots])

    def _simulate(self):
        if any(getattr(lane, 'primed', False) for lane in self.lanes):
            # a lane primed by play_move_pipelined has the coming move's simulations in flight already: searching it again
            # would double its visits -- the lanes are taken one by one, the primed ones as they are
            for lane in self.lanes:
                if not getattr(lane, 'primed', False):
                    self._simulate_lane(lane)
                lane.primed = False
            return
        n = self.eng.n_playout
        if all(getattr(lane.evaluator, 'resident_ok', None) is not None and lane.evaluator.resident_ok(lane.eng) for lane in self.lanes):
            for lane in self.lanes:   # one launch per lane for the whole search: nothing to interleave, no graph
                self._simulate_lane(lane)
            return
        if self.use_graph:
            per = self.eng.graph_chunk(self.sims_per_graph)
            full, n = divmod(n, per)
            for _ in range(full):
                # timing samples: every k-th chunk of the run (counted across moves) is launched eagerly
                c = self._chunks_done = getattr(self, '_chunks_done', -1) + 1
                eager = self.eager_every > 0 and c % self.eager_every == 0
                for lane in self.lanes:
                    with self._on(lane):
                        if eager:
                            lane.eng.sim_chunk(lane.evaluator, per)
                        else:
                            lane.eng.simulate(lane.evaluator, per, use_graph=True, sims_per_graph=per)
        if n and len(self.lanes) == 1:
            lane = self.lanes[0]
            with self._on(lane):
                lane.eng.sim_chunk(lane.evaluator, n)
        elif n:
            # interleave the lanes in chunks so that their kernels alternate on the device
            chunk = self.eng.graph_chunk(8)
            for c0 in range(0, n, chunk):
                for lane in self.lanes:
                    with self._on(lane):
                        lane.eng.sim_chunk(lane.evaluator, min(chunk, n - c0))

    def _simulate_lane(self, lane):
        """Enqueue the n_playout simulations of ONE lane on its stream (graph replays + the eager remainder)."""
        n = self.eng.n_playout
        with self._on(lane):
            res_ok = getattr(lane.evaluator, 'resident_ok', None)
            if res_ok is not None and res_ok(lane.eng):   # one launch for the whole search: no graph, no chunks
                lane.eng.sim_chunk(lane.evaluator, n)
                return
            if self.use_graph:
                per = self.eng.graph_chunk(self.sims_per_graph)
                full, n = divmod(n, per)
                for _ in range(full):
                    c = self._chunks_done = getattr(self, '_chunks_done', -1) + 1
                    if self.eager_every > 0 and c % self.eager_every == 0:
                        lane.eng.sim_chunk(lane.evaluator, per)
                    else:
                        lane.eng.simulate(lane.evaluator, per, use_graph=True, sims_per_graph=per)
            if n:
                lane.eng.sim_chunk(lane.evaluator, n)

    # -- one move for every running game ----------------------------------------------
    def _finish_lane(self, lane):
        """The move of one lane after its simulations: root visits -> pi -> move drawn (alphazero_mcts.py:88-92,148),
        tree reuse, game step.  Synchronises that lane's stream only.  Returns the trajectories that ended."""
        eng = lane.eng
        lo = lane.slots.start
        running = lo + np.nonzero(self.slot_game[lane.slots] >= 0)[0]
        with self._on(lane):
            visits = lane.eng.root_visits()
            # the split-f16 trunk reports an input outside its range: float planes only (positions are 0 / 1 planes by construction,
            # and the query waits for the device)
            if hasattr(getattr(lane.evaluator, 'hip', None), 'check_flags') and getattr(lane.evaluator, 'needs_obs', True):
                lane.evaluator.hip.check_flags()
        self.sims_done += eng.n_playout * len(running)
        moves = np.full(eng.n_games, -2, dtype=np.int32)
        if len(running):
            us = move_uniform(self.seed, self.slot_game[running], self.slot_ply[running])
            taken = self.cell_taken[running]
            if eng.game == 'connect4':  # action = column, legal while its top cell is empty; the stone drops
                legal = ~taken[:, (eng.rows - 1) * eng.cols:]
                heights = taken.reshape(len(running), eng.rows, eng.cols).sum(axis=1)
            else:
                legal = ~taken
            pis, chosen = batch_pi_and_moves(visits[running - lo], legal, self.temperature, us)
            cells = heights[np.arange(len(running)), chosen] * eng.cols + chosen if eng.game == 'connect4' else chosen
            moves[running - lo] = chosen
            self.cell_taken[running, cells] = True
            self.slot_ply[running] += 1
            for i, s_ in enumerate(running):
                self.slot_pis[s_].append(pis[i])
                self.slot_moves[s_].append(int(chosen[i]))
        step_moves = np.where(moves >= 0, moves, -1).astype(np.int32)
        with self._on(lane):
            winner, ended = lane.eng.advance_and_step(moves, step_moves)  # tree reuse before the boards change
        self.moves_done += len(running)
        done = []
        for s in running:
            if ended[s - lo]:
                done.append(Trajectory(self.slot_game[s], eng.board_size, eng.n_in_row,
                                       self.slot_moves[s], self.slot_pis[s], winner[s - lo], game=eng.game))
                self.slot_game[s] = -1
        return done

    def play_move(self):
        """n_playout simulations, then pick / apply one move per game.  Returns the list of
        trajectories of the games that ended with this move."""
        self._simulate()
        done = []
        for lane in self.lanes:
            done.extend(self._finish_lane(lane))
        return done

    def play_move_pipelined(self, refill=None):
        """play_move() with the host side of one lane's move hidden under the other lanes' simulations: per lane, wait
        for ITS simulations, draw and apply its moves, refill its finished slots (``refill(n) -> up to n new game
        ids``, optional) and enqueue its NEXT move's simulations at once -- while a lane is on the host the others keep
        the GPU busy.  Every call completes exactly one move of every running game, like play_move(); between calls
        the simulations of the coming move are already in flight.  Results are those of play_move(): games are
        independent and their uniforms are keyed by (game, ply)."""
        for lane in self.lanes:
            if not getattr(lane, 'primed', False) and (self.slot_game[lane.slots] >= 0).any():
                self._simulate_lane(lane)
                lane.primed = True
        done_all = []
        for lane in self.lanes:
            if not getattr(lane, 'primed', False):
                continue
            done = self._finish_lane(lane)
            lane.primed = False
            if done:
                if refill is not None:
                    free = lane.slots.start + np.nonzero(self.slot_game[lane.slots] < 0)[0]
                    ids = list(refill(len(free)))[:len(free)]
                    if ids:
                        self._start(free[:len(ids)], ids)
                self._retire_lane(lane)
            if (self.slot_game[lane.slots] >= 0).any():
                self._simulate_lane(lane)
                lane.primed = True
            done_all.extend(done)
        return done_all

    def _retire_lane(self, lane):
        idle = np.full(lane.eng.n_games, -2, dtype=np.int32)
        idle[np.nonzero(self.slot_game[lane.slots] < 0)[0]] = -1
        with self._on(lane):
            lane.eng.advance(idle)
            lane.eng.set_active((self.slot_game[lane.slots] >= 0).astype(np.uint8))

    def retire_finished(self):
        """reset_player(): discard the trees of idle slots (game.py:128) and mask them out."""
        for lane in self.lanes:
            self._retire_lane(lane)

    def abandon_running(self):
        """Drop every game still in a slot (a run that stopped early -- max_moves -- leaves its games there, and a pipelined one
        leaves their NEXT search finished on the device as well): the slots become idle, their trees are reset, no lane stays
        primed.  run() starts with this: games of an earlier run are neither played on nor searched a second time on top of a
        finished search."""
        if not (self.slot_game >= 0).any() and not any(getattr(lane, 'primed', False) for lane in self.lanes):
            return
        self.slot_game[:] = -1
        for lane in self.lanes:
            if getattr(lane, 'primed', False):
                lane.stream.synchronize()
                lane.primed = False
            self._retire_lane(lane)

    def check(self):
        return [lane.eng.check() for lane in self.lanes]

    def run(self, game_ids, max_moves=None, pipelined=False):
        """Play all ``game_ids`` to the end; returns trajectories sorted by game id.  ``pipelined``: the host side of a
        lane's move under the other lanes' simulations (play_move_pipelined); same trajectories."""
        self.abandon_running()
        if pipelined:
            return self._run_pipelined(game_ids, max_moves)
        pending = list(game_ids)
        G = self.n_slots
        first = pending[:G]
        pending = pending[G:]
        self._start(range(len(first)), first)
        self._set_active()
        out = []
        n_moves = 0
        while (self.slot_game >= 0).any():
            done = self.play_move()
            out.extend(done)
            n_moves += 1
            free = np.nonzero(self.slot_game < 0)[0]
            if done and pending:
                take = pending[:len(free)]
                pending = pending[len(take):]
                self._start(free[:len(take)], take)
            if done:
                self.retire_finished()
            if max_moves is not None and n_moves >= max_moves:
                break
        self.check()
        return sorted(out, key=lambda t: t.game_id)


    def _run_pipelined(self, game_ids, max_moves=None):
        pending = list(game_ids)
        first, pending = pending[:self.n_slots], pending[self.n_slots:]
        self._start(range(len(first)), first)
        self._set_active()

        def refill(n):
            take = pending[:n]
            del pending[:n]
            return take

        out, n_moves = [], 0
        while (self.slot_game >= 0).any():
            out.extend(self.play_move_pipelined(refill))
            n_moves += 1
            if max_moves is not None and n_moves >= max_moves:
                break
        self.torch.cuda.synchronize()   # (lanes still primed hold a finished search: play_move() / play_move_pipelined() use it)
        self.check()
        return sorted(out, key=lambda t: t.game_id)


    # -- the move step on the device (include/rlzero_hip.h: rz_play_*) -----------------------------------------------
    # The host enqueues whole moves -- search, draw, priors, tree reuse, game step, end / refill of slots -- and never waits for
    # one: what happened comes back through each lane's log, read behind the GPU.  pi is formed HERE, from the logged visit
    # counts with the reference's numpy expression (alphazero_mcts.py:10-14,91-92), and every move the device drew is checked
    # against numpy's inverse-CDF draw on the same uniform (a mismatch raises; a draw too close to an interval edge was never
    # made by the device: the slot stalls until the move computed here is handed back).  Same trajectories as play_move().
    def device_attach(self, queue_capacity=1 << 16, ring_steps=64, depth=None, copy_every=None, stall_margin=0.0, move_graphs=True):
        """Switch this object to device-driven moves.  ``copy_every``: moves of a lane per read-back of its log rows (None: 1, or
        8 when a move is a fraction of a millisecond -- boards of a few cells with few simulations); ``depth``: read-backs a lane
        may be ahead of the rows processed here (None: 1 when a move is long, else 2).  ``move_graphs``: a lane whose search is the
        one-launch resident kernel replays its WHOLE move -- selection, search, draw, priors, tree reuse, game step, refill -- from
        one hipGraph."""
        t = self.torch
        from ._hip import PLAY_RECORD_WORDS
        dev = self.eng.device
        if getattr(self, '_dev_on', False):
            self.device_stop()
        self._attach_kw = dict(ring_steps=ring_steps, depth=depth, copy_every=copy_every, stall_margin=stall_margin, move_graphs=move_graphs)
        t.cuda.synchronize(dev)
        work = self.eng.n_playout * self.eng.n_cells   # ~ the length of a move
        self._copy_every = int(copy_every) if copy_every else (1 if work >= 4000 else 8)
        self._depth = int(depth) if depth else (1 if work >= 100000 else 2)
        ring_steps = max(int(ring_steps), self._copy_every * (self._depth + 2) + 2)
        self._queue_ids = t.zeros(int(queue_capacity), dtype=t.int64, device=dev)
        self._queue_ctl = t.zeros(2, dtype=t.int32, device=dev)
        self._queue_len, self._started = 0, 0
        words = PLAY_RECORD_WORDS + self.eng.n_actions
        for lane in self.lanes:
            with self._on(lane):
                lane.eng.play_attach(self.seed, self.temperature, self._queue_ids, self._queue_ctl, ring_steps=ring_steps,
                                     stall_margin=stall_margin)
                lane.move_graph = lane.eng.warm_move_graph(lane.evaluator) if move_graphs else None
            # (the engine's log ring is pinned host memory that its kernels write directly: nothing to copy -- or, RZ_PLAY_DEVICE_LOG=1,
            # a device ring whose rows _read_back copies)
            lane.host_log = lane.eng.play_log if lane.eng.play_log_on_host else t.empty((int(ring_steps), lane.eng.n_games, words), dtype=t.int32, pin_memory=True)
            lane.host_np = lane.host_log.numpy()
            lane.uncopied = []          # rows written by enqueued moves, their read-back not enqueued yet
            lane.inflight = []          # [(rows, event)] read-backs enqueued, oldest first
            lane.last_running = -1      # RUNNING records in the last row read (-1: none read yet)
            lane.primed = False
        max_plies = self.eng.n_cells
        self._pi_buf = np.empty((self.n_slots, max_plies, self.eng.n_actions), dtype=np.float64)   # (pages are touched as games grow)
        self._mv_buf = np.zeros((self.n_slots, max_plies), dtype=np.int32)
        self._stalls = {}               # slot -> (game id, ply, pi, move): decided here, waiting for the device to take it
        self.stalls_resolved = 0
        self.slot_game[:] = -1
        self._dev_on = True
        t.cuda.synchronize(dev)

    def device_queue(self, game_ids):
        """Replace the queue of waiting game ids (synchronises: not for the middle of a run) and let idle slots take from it."""
        t = self.torch
        ids = np.ascontiguousarray(list(game_ids), dtype=np.int64)
        if ids.size > self._queue_ids.numel():
            raise ValueError('%d game ids for a queue of %d: device_attach(queue_capacity=...)' % (ids.size, self._queue_ids.numel()))
        t.cuda.synchronize(self.eng.device)
        self._queue_ids[:ids.size].copy_(t.from_numpy(ids))
        self._queue_ctl.copy_(t.tensor([0, ids.size], dtype=t.int32))
        t.cuda.synchronize(self.eng.device)
        self._queue_len, self._started = int(ids.size), 0
        for lane in self.lanes:
            with self._on(lane):
                lane.eng.play_refill()
            lane.last_running = -1

    def _lane_quiet(self, lane):
        """Nothing left to do on this lane, as far as the rows read so far can tell (the host's view lags the device by design)."""
        return not lane.inflight and not lane.uncopied and lane.last_running == 0 and self._started >= self._queue_len

    def _read_back(self, lane):
        """An event behind the lane's new log rows (they are in pinned memory when it has passed: written there by the kernels, or
        -- a device ring -- copied there, contiguous runs of the ring in one copy each)."""
        rows = lane.uncopied
        if not rows:
            return
        lane.uncopied = []
        with self._on(lane):
            at = 0
            while at < len(rows):
                end = at + 1
                while end < len(rows) and rows[end] == rows[end - 1] + 1:
                    end += 1
                if not lane.eng.play_log_on_host:
                    lane.host_log[rows[at]:rows[end - 1] + 1].copy_(lane.eng.play_log[rows[at]:rows[end - 1] + 1], non_blocking=True)
                at = end
            ev = self.torch.cuda.Event()
            ev.record(lane.stream)
        lane.inflight.append((rows, ev))

    def play_move_device(self):
        """Enqueue ONE more move of every lane that may still have games and process the log rows that have arrived
        -> the trajectories of the games found finished in them."""
        for lane in self.lanes:
            if self._lane_quiet(lane):
                continue
            if lane.move_graph is not None:
                with self._on(lane):
                    lane.uncopied.append(lane.eng.play_move_replay(lane.move_graph))
            else:
                self._simulate_lane(lane)
                with self._on(lane):
                    lane.uncopied.append(lane.eng.play_move())
            if len(lane.uncopied) >= self._copy_every:
                self._read_back(lane)
        done = []
        for lane in self.lanes:
            done.extend(self._harvest(lane, keep=self._depth))
        return done

    def device_drain(self):
        """Wait for every enqueued move and process its row -> the finished trajectories found."""
        done = []
        for lane in self.lanes:
            self._read_back(lane)
        for lane in self.lanes:
            done.extend(self._harvest(lane, keep=0))
        return done

    def _harvest(self, lane, keep):
        from ._hip import PLAY_ENDED, PLAY_RECORD_WORDS, PLAY_RESOLVED, PLAY_RUNNING, PLAY_SEARCHED, PLAY_STALLED, HipError
        rows = []
        while lane.inflight and (len(lane.inflight) > keep or lane.inflight[0][1].query()):
            batch, ev = lane.inflight.pop(0)
            ev.synchronize()
            rows.extend(batch)
        if not rows:
            return []
        eng, lo, W0 = lane.eng, lane.slots.start, PLAY_RECORD_WORDS
        G = eng.n_games
        rec = lane.host_np[rows]                                # [rows, G, words] (a copy: the pinned rows may be overwritten from now on)
        flags = rec[:, :, 4] & 0xFFFF
        lane.last_running = int(((flags[-1] & PLAY_RUNNING) != 0).sum())
        row_i, g_i = np.nonzero(flags & PLAY_RUNNING)           # (row-major: a slot's records in move order)
        if row_i.size == 0:
            return []
        rec, flags = rec[row_i, g_i], flags[row_i, g_i]
        slots = lo + g_i
        gids = rec[:, 0].astype(np.uint32).astype(np.int64) | (rec[:, 1].astype(np.int64) << 32)
        plies, moves = rec[:, 2].astype(np.int64), rec[:, 3]
        visits = rec[:, W0:]
        legal = visits >= 0
        # the reference's expression on the logged counts; the draw with the game's uniform (numpy's inverse-CDF rule): the arbiter
        pis, chosen = batch_pi_and_moves(np.where(legal, visits, 0), legal, self.temperature, move_uniform(self.seed, gids, plies))
        not_plain = (flags & (PLAY_STALLED | PLAY_RESOLVED)) != 0
        wrong = ~not_plain & (chosen != moves)
        if wrong.any():
            bad = np.nonzero(wrong)[0][0]
            raise HipError('the move drawn on the device (%d) is not numpy\'s (%d): game %d, ply %d' % (moves[bad], chosen[bad], gids[bad], plies[bad]))
        self.sims_done += eng.n_playout * int(((flags & PLAY_SEARCHED) != 0).sum())
        special = not_plain | ((flags & PLAY_ENDED) != 0) | (plies == 0)
        done = []
        first = np.searchsorted(row_i, np.arange(len(rows) + 1))   # records of row r: first[r] .. first[r + 1]
        for r in range(len(rows)):
            a, b = int(first[r]), int(first[r + 1])
            if a == b:
                continue
            easy = np.nonzero(~special[a:b])[0] + a
            if easy.size:   # a move in the middle of a game: the whole row at once
                s = slots[easy]
                if (self.slot_game[s] != gids[easy]).any() or (self.slot_ply[s] != plies[easy]).any():
                    i = easy[np.nonzero((self.slot_game[s] != gids[easy]) | (self.slot_ply[s] != plies[easy]))[0][0]]
                    raise HipError('slot %d: the log says game %d ply %d, the host expected game %d ply %d' % (
                        slots[i], gids[i], plies[i], self.slot_game[slots[i]], self.slot_ply[slots[i]]))
                self._pi_buf[s, plies[easy]] = pis[easy]
                self._mv_buf[s, plies[easy]] = moves[easy]
                self.slot_ply[s] += 1
                self.moves_done += int(easy.size)
            for i in np.nonzero(special[a:b])[0] + a:
                s, f, gid, ply = int(slots[i]), int(flags[i]), int(gids[i]), int(plies[i])
                if f & PLAY_STALLED:
                    known = self._stalls.get(s)
                    if known is None or known[:2] != (gid, ply):   # first sight of this stall: decide, hand the move back
                        self._stalls[s] = (gid, ply, pis[i], int(chosen[i]))
                        with self._on(lane):
                            eng.play_resolve(s - lo, int(chosen[i]))
                    continue
                pi, mv = pis[i], int(moves[i])
                if f & PLAY_RESOLVED:
                    known = self._stalls.pop(s, None)
                    if known is None or known[:2] != (gid, ply) or known[3] != mv:
                        raise HipError('slot %d: the device resolved game %d ply %d with move %d, the host had decided %r' % (s, gid, ply, mv, known))
                    pi = known[2]
                    self.stalls_resolved += 1
                if ply == 0:   # the slot has started this game
                    self.slot_game[s], self.slot_ply[s] = gid, 0
                    self._started += 1
                if self.slot_game[s] != gid or self.slot_ply[s] != ply:
                    raise HipError('slot %d: the log says game %d ply %d, the host expected game %d ply %d' % (s, gid, ply, self.slot_game[s], self.slot_ply[s]))
                self._pi_buf[s, ply] = pi
                self._mv_buf[s, ply] = mv
                self.slot_ply[s] += 1
                self.moves_done += 1
                if f & PLAY_ENDED:
                    winner = ((int(rec[i, 4]) >> 16) & 3) - 1
                    n = ply + 1
                    done.append(Trajectory(gid, eng.board_size, eng.n_in_row, self._mv_buf[s, :n].tolist(), self._pi_buf[s, :n].copy(), winner, game=eng.game))
                    self.slot_game[s] = -1
        return done

    def device_stop(self):
        """Drop every game and every row in flight: all slots idle (a run that stops early)."""
        for lane in self.lanes:
            lane.stream.synchronize()
            with self._on(lane):
                lane.eng.play_stop()
            lane.stream.synchronize()
            lane.inflight, lane.uncopied, lane.last_running, lane.primed = [], [], 0, False
        self.slot_game[:] = -1
        self._stalls = {}
        self._queue_len = self._started = 0

    def run_device(self, game_ids, max_moves=None, on_finished=None):
        """run() with the move step on the device: same trajectories, sorted by game id.  ``on_finished(trajectories)``: called with the
        games found finished after every enqueued move, while the GPU searches on (a consumer's per-game work -- the trainer's
        observation planes and replay-buffer entries -- then costs the round nothing)."""
        game_ids = list(game_ids)
        if not getattr(self, '_dev_on', False):
            self.device_attach(queue_capacity=max(len(game_ids), 1))
        elif len(game_ids) > self._queue_ids.numel():
            self.device_attach(queue_capacity=len(game_ids), **self._attach_kw)   # (a longer queue: attach again, same settings)
        else:
            self.device_stop()
        self.device_queue(game_ids)
        out, n_moves = [], 0
        while len(out) < len(game_ids):
            if all(self._lane_quiet(lane) for lane in self.lanes):
                raise RuntimeError('device-driven self-play went quiet with %d of %d games finished' % (len(out), len(game_ids)))
            done = self.play_move_device()
            out.extend(done)
            if on_finished is not None and done:
                on_finished(done)
            n_moves += 1
            if max_moves is not None and n_moves >= max_moves:
                break
        done = self.device_drain()
        out.extend(done)
        if on_finished is not None and done:
            on_finished(done)
        self.check()
        if len(out) < len(game_ids):
            self.device_stop()
        return sorted(out, key=lambda t: t.game_id)


# ------------------------------------------------------------------------- multi-GPU gather
def pack_trajectories(trajs, n_cells):
    """-> (header int64 [n,4] = game id, plies, winner, 0 ; moves int64 [P] ; pis float64 [P,S])."""
    header = np.array([[t.game_id, len(t.moves), t.winner, 0] for t in trajs], dtype=np.int64).reshape(-1, 4)
    moves = np.array([m for t in trajs for m in t.moves], dtype=np.int64)
    pis = np.concatenate([t.pis for t in trajs], axis=0) if trajs else np.zeros((0, n_cells))
    return header, moves, pis.reshape(-1, n_cells)


def unpack_trajectories(header, moves, pis, board_size, n_in_row, game='gomoku'):
    out, at = [], 0
    for gid, plies, winner, _ in header:
        plies = int(plies)
        out.append(Trajectory(gid, board_size, n_in_row, moves[at:at + plies], pis[at:at + plies], winner,
                              game=game))
        at += plies
    return out


COLLECTIVES_PER_EXCHANGE = 2   # gather_trajectories: one all_gather of a 3-word size row + ONE gather of the byte payload


def _payload_bytes(header, moves, pis, pi_dtype):
    """One contiguous record of a rank's finished games: header int64 [n, 4] | moves int64 [P] | pi [P, A] (float32 or
    float64) -- fixed strides, so (n, P) from the size row are all the receiver needs to cut it up again."""
    return np.concatenate([np.ascontiguousarray(header, dtype=np.int64).reshape(-1).view(np.uint8),
                           np.ascontiguousarray(moves, dtype=np.int64).view(np.uint8),
                           np.ascontiguousarray(pis, dtype=pi_dtype).reshape(-1).view(np.uint8)])


def payload_of(trajs, n_cells, pi_dtype):
    """_payload_bytes(*pack_trajectories(trajs), pi_dtype) in ONE pass: every game's pi is converted straight into its place in the
    buffer (a collection round's pi is 100+ MB per rank: packing it through three intermediate copies cost more than sending it).
    -> (bytes uint8, games, plies)."""
    n_games, n_plies = len(trajs), sum(len(t.moves) for t in trajs)
    item = np.dtype(pi_dtype).itemsize
    raw = np.empty(n_games * 32 + n_plies * (8 + n_cells * item), dtype=np.uint8)
    header = raw[:n_games * 32].view(np.int64).reshape(n_games, 4)
    moves = raw[n_games * 32:n_games * 32 + 8 * n_plies].view(np.int64)
    pis = raw[n_games * 32 + 8 * n_plies:].view(pi_dtype).reshape(n_plies, n_cells)
    at = 0
    for i, t in enumerate(trajs):
        k = len(t.moves)
        header[i] = (t.game_id, k, t.winner, 0)
        moves[at:at + k] = t.moves
        if k:
            pis[at:at + k] = t.pis   # (numpy converts while it copies)
        at += k
    return raw, n_games, n_plies


def _payload_split(raw, n_games, n_plies, n_cells, pi_dtype):
    """-> header, moves, pi as VIEWS of the received bytes (float32 pi stays float32: Trajectory keeps it)."""
    at = n_games * 32
    header = raw[:at].view(np.int64).reshape(n_games, 4)
    moves = raw[at:at + 8 * n_plies].view(np.int64)
    at += 8 * n_plies
    pis = raw[at:at + n_plies * n_cells * np.dtype(pi_dtype).itemsize].view(pi_dtype).reshape(n_plies, n_cells)
    return header, moves, pis


def gather_trajectories(trajs, board_size, n_in_row, dst=0, group=None, game='gomoku', pi_dtype=np.float64):
    """The one exchange of the path: every rank sends its finished trajectories to ``dst``.  Two collectives
    (COLLECTIVES_PER_EXCHANGE): an all_gather of one 3-word row per rank (games, plies, error bit) and ONE gather of a byte
    payload -- header, moves and pi of all the rank's games in one buffer, padded to the longest rank's.  Returns the merged,
    game-id-sorted list on ``dst`` and None elsewhere.  Without an initialised process group: identity.
    ``pi_dtype``: what pi travels as -- float64 (default: bit-identical to the single-process run) or float32 (what the
    learner consumes, alphazero_agent.py:59-61: half the bytes, the exchange SURVEY.md 8e sizes; the trainer's choice).

    What can fail locally -- packing, the device copy of the rank's own payload -- happens BEFORE the size row is exchanged
    and travels in it as the error bit: a rank that failed still takes part in that all_gather and then ALL ranks raise, so
    no peer is left blocked in the gather.  (Behind the size row only the padding of the send buffer and the receive buffers
    of ``dst`` are allocated; running out of memory there is an out-of-memory inside a collective, fatal to the job.)"""
    import torch
    import torch.distributed as dist
    n_cells = board_size[1] if game == 'connect4' else board_size * board_size  # width of a pi row
    if not (dist.is_available() and dist.is_initialized()):
        return sorted(trajs, key=lambda t: t.game_id)
    rank, world = dist.get_rank(group), dist.get_world_size(group)  # a group of one runs the same collectives
    on_gpu = dist.get_backend(group) == 'nccl'
    device = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    pi_dtype = np.dtype(pi_dtype).type
    local_error, mine, n_games, n_plies = None, None, 0, 0
    try:
        payload, n_games, n_plies = payload_of(trajs, n_cells, pi_dtype)
        mine = torch.from_numpy(payload).to(device)
    except Exception as exc:  # noqa: BLE001
        local_error, mine, n_games, n_plies = exc, None, 0, 0
    sizes = torch.tensor([n_games, n_plies, 0 if local_error is None else 1], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)                                   # collective 1 of 2
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    if all_sizes[:, 2].any():
        raise RuntimeError('gather_trajectories: packing failed on rank(s) %s%s' % (
            np.nonzero(all_sizes[:, 2])[0].tolist(), '' if local_error is None else ' (here: %r)' % (local_error, )))
    row_bytes = 8 + n_cells * np.dtype(pi_dtype).itemsize
    n_bytes = all_sizes[:, 0] * 32 + all_sizes[:, 1] * row_bytes
    send = torch.empty(max(int(n_bytes.max()), 1), dtype=torch.uint8, device=device)   # (padded to the longest rank's; the padding is never read)
    send[:mine.numel()] = mine
    bucket = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, gather_list=bucket, dst=dst, group=group)                      # collective 2 of 2
    if rank != dst:
        return None
    merged = []
    for r in range(world):
        raw = bucket[r][:int(n_bytes[r])].cpu().numpy()
        merged.extend(unpack_trajectories(*_payload_split(raw, int(all_sizes[r, 0]), int(all_sizes[r, 1]), n_cells, pi_dtype),
                                          board_size, n_in_row, game=game))
    return sorted(merged, key=lambda t: t.game_id)


def broadcast_weights(module, src=0, group=None):
    """After ``policy_update`` on rank ``src``: one broadcast of the flattened parameters
    (1.3 MB at 15x15) so every rank's self-play uses the same network.  RCCL when the process
    group backend is nccl; identity without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return module
    params = list(module.parameters())
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        if dist.get_backend(group) == 'nccl' and not flat.is_cuda:
            flat = flat.cuda()
        dist.broadcast(flat, src=src, group=group)
        at = 0
        for p in params:
            n = p.numel()
            # copy_ on the Parameter ITSELF (not on p.data, which carries a version counter of its own): the in-place write bumps
            # p._version, which is what HipNetEvaluator.refresh_if_changed() looks at -- a rank that only receives weights must
            # re-upload them like the rank whose optimiser stepped
            p.copy_(flat[at:at + n].reshape(p.shape).to(p.device))
            at += n
    return module
