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


def shard_game_ids(n_games_total, rank, world_size):
    return list(range(rank, n_games_total, world_size))


class Trajectory(object):
    """One finished game: what start_self_play returns, in compact form."""

    def __init__(self, game_id, board_size, n_in_row, moves, pis, winner):
        self.game_id = int(game_id)
        self.board_size, self.n_in_row = int(board_size), int(n_in_row)
        self.moves = [int(m) for m in moves]
        self.pis = np.asarray(pis, dtype=np.float64).reshape(len(self.moves), board_size * board_size)
        self.winner = int(winner)

    def z(self):
        """+1 for the plies of the winner, -1 for the loser's, 0 on a tie (game.py:121-126)."""
        movers = np.arange(len(self.moves)) % 2  # player 0 moves first
        if self.winner == -1:
            return np.zeros(len(self.moves))
        return np.where(movers == self.winner, 1.0, -1.0)

    def states(self):
        """Observation planes before every move (GomokuEnv.current_state)."""
        from .games.gomoku.gomoku_env import GomokuEnv
        env = GomokuEnv(self.board_size, self.n_in_row)
        env.reset()
        out = []
        for m in self.moves:
            out.append(env.current_state())
            env.step(m)
        return out

    def as_reference_tuple(self):
        """(winner, [(state, mcts_prob, z), ...]) -- start_self_play's return value."""
        return self.winner, list(zip(self.states(), list(self.pis), self.z()))


class BatchedSelfPlay(object):
    """Plays ``len(game_ids)`` games on ``engine`` (slots are refilled as games end)."""

    def __init__(self, engine, evaluator, temperature=1.0, seed=0, use_graph=False,
                 sims_per_graph=8):
        self.eng = engine
        self.evaluator = evaluator
        self.temperature = float(temperature)
        self.seed = int(seed)
        self.use_graph = use_graph
        self.sims_per_graph = sims_per_graph
        G = engine.n_games
        self.slot_game = np.full(G, -1, dtype=np.int64)
        self.slot_ply = np.zeros(G, dtype=np.int64)
        self.slot_occ = [0] * G
        self.slot_moves = [[] for _ in range(G)]
        self.slot_pis = [[] for _ in range(G)]
        self.sims_done = 0
        self.moves_done = 0

    # -- slot management -----------------------------------------------------------
    def _start(self, slots, game_ids):
        mask = np.zeros(self.eng.n_games, dtype=np.uint8)
        for s, g in zip(slots, game_ids):
            mask[s] = 1
            self.slot_game[s] = g
            self.slot_ply[s] = 0
            self.slot_occ[s] = 0
            self.slot_moves[s] = []
            self.slot_pis[s] = []
        self.eng.reset_games(mask=mask)

    def _set_active(self):
        self.eng.set_active((self.slot_game >= 0).astype(np.uint8))

    # -- one move for every running game ----------------------------------------------
    def play_move(self):
        """n_playout simulations, then pick / apply one move per game.  Returns the list of
        trajectories of the games that ended with this move."""
        eng = self.eng
        S = eng.n_cells
        running = np.nonzero(self.slot_game >= 0)[0]
        eng.simulate(self.evaluator, eng.n_playout, use_graph=self.use_graph,
                     sims_per_graph=self.sims_per_graph)
        visits = eng.root_visits()
        self.sims_done += eng.n_playout * len(running)
        moves = np.full(eng.n_games, -2, dtype=np.int32)
        us = move_uniform(self.seed, self.slot_game[running], self.slot_ply[running])
        for s, u in zip(running, us):
            occ = self.slot_occ[s]
            acts = np.array([c for c in range(S) if not (occ >> c) & 1])
            probs = visits_to_pi(visits[s, acts], self.temperature)
            pi = np.zeros(S)
            pi[acts] = probs
            move = draw_move(acts, probs, u)
            moves[s] = move
            self.slot_pis[s].append(pi)
            self.slot_moves[s].append(move)
            self.slot_occ[s] = occ | (1 << move)
            self.slot_ply[s] += 1
        eng.advance(moves)  # tree reuse (update_with_move), before the boards change
        step_moves = np.where(moves >= 0, moves, -1).astype(np.int32)
        winner, ended = eng.step(step_moves)
        self.moves_done += len(running)
        done = []
        for s in running:
            if ended[s]:
                done.append(Trajectory(self.slot_game[s], eng.board_size, eng.n_in_row,
                                       self.slot_moves[s], self.slot_pis[s], winner[s]))
                self.slot_game[s] = -1
        return done

    def run(self, game_ids, max_moves=None):
        """Play all ``game_ids`` to the end; returns trajectories sorted by game id."""
        pending = list(game_ids)
        G = self.eng.n_games
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
                # reset_player(): the trees of finished games are discarded (game.py:128)
                idle = np.full(G, -2, dtype=np.int32)
                idle[np.nonzero(self.slot_game < 0)[0]] = -1
                self.eng.advance(idle)
                self._set_active()
            if max_moves is not None and n_moves >= max_moves:
                break
        self.eng.check()
        return sorted(out, key=lambda t: t.game_id)


# ------------------------------------------------------------------------- multi-GPU gather
def pack_trajectories(trajs, n_cells):
    """-> (header int64 [n,4] = game id, plies, winner, 0 ; moves int64 [P] ; pis float64 [P,S])."""
    header = np.array([[t.game_id, len(t.moves), t.winner, 0] for t in trajs], dtype=np.int64).reshape(-1, 4)
    moves = np.array([m for t in trajs for m in t.moves], dtype=np.int64)
    pis = np.concatenate([t.pis for t in trajs], axis=0) if trajs else np.zeros((0, n_cells))
    return header, moves, pis.reshape(-1, n_cells)


def unpack_trajectories(header, moves, pis, board_size, n_in_row):
    out, at = [], 0
    for gid, plies, winner, _ in header:
        plies = int(plies)
        out.append(Trajectory(gid, board_size, n_in_row, moves[at:at + plies], pis[at:at + plies], winner))
        at += plies
    return out


def gather_trajectories(trajs, board_size, n_in_row, dst=0, group=None):
    """The one exchange of the path: every rank sends its finished trajectories to ``dst``
    (one size all_gather + one padded gather per array).  Returns the merged, game-id-sorted
    list on ``dst`` and None elsewhere.  Without an initialised process group: identity."""
    import torch
    import torch.distributed as dist
    n_cells = board_size * board_size
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sorted(trajs, key=lambda t: t.game_id)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == 'nccl'
    device = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    header, moves, pis = pack_trajectories(trajs, n_cells)
    sizes = torch.tensor([header.shape[0], moves.shape[0]], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    max_games, max_plies = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())

    def padded(arr, rows, dtype):
        buf = torch.zeros((rows, ) + arr.shape[1:], dtype=dtype, device=device)
        if arr.shape[0]:
            buf[:arr.shape[0]] = torch.from_numpy(arr).to(device)
        return buf

    sends = (padded(header, max_games, torch.int64), padded(moves, max_plies, torch.int64),
             padded(pis, max_plies, torch.float64))
    recvs = []
    for buf in sends:
        bucket = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, gather_list=bucket, dst=dst, group=group)
        recvs.append(bucket)
    if rank != dst:
        return None
    merged = []
    for r in range(world):
        n_g, n_p = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        merged.extend(unpack_trajectories(recvs[0][r][:n_g].cpu().numpy(), recvs[1][r][:n_p].cpu().numpy(),
                                          recvs[2][r][:n_p].cpu().numpy(), board_size, n_in_row))
    return sorted(merged, key=lambda t: t.game_id)
