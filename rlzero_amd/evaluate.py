"""Evaluation games in lock-step on one GPU: the current policy against the pure-MCTS opponent.

Batched counterpart of the loop in ``TrainPipeline.policy_evaluate`` (tools/train_alphazero.py:139-162): every game is one
``GameControl.start_play(AlphaZeroPlayer, RolloutPlayer)`` (rlzero/games/gomoku/game.py:61-94) and does, ply by ply, exactly what
the two players of the reference do for one game --

* the network player (``AlphaZeroPlayer(is_selfplay=False)``, rlzero/mcts/alphazero_mcts.py:136-165): n_playout simulations from a
  FRESH root without noise, pi = softmax(log(N + 1e-10) / T) with the evaluation temperature 1e-3, TWO draws from pi of which the
  second is played (:148,157), root reset (:158);
* the pure-MCTS player (``RolloutPlayer``, rlzero/mcts/rollout_mcts.py:110-140): n_playout simulations with uniform priors and a
  random play-out as the leaf value (:49-74), the most visited move -- first maximum in ascending action order (:88-94) --, root reset;

-- but the searches of all games advance together.  Two engines hold the same boards: one searches with the hand-written network
evaluator, the other with the device play-outs; at every ply each engine searches the games whose mover it plays (its ``active``
mask), on its own HIP stream, and both apply every move.  The reference seats the network player as player 0 and ``reset()`` ignores
``start_player`` (game.py:69-76, gomoku_env.py:33-47), so by default the network moves first in every game; ``net_first`` lets a
caller seat it second in some games, in which case the two engines search their halves of the batch at the same time.

Randomness: the two draws of a network move use uniforms keyed (seed, game, 2 ply) and (seed, game, 2 ply + 1) (selfplay.move_uniform);
the play-outs of a pure-MCTS move use the device generator keyed (rollout_seed(seed, ply) ^ slot, simulation, play-out ply) -- slot =
the game's position in ``run(game_ids)``, the game id itself by default --, which is what a single-game ``RolloutPlayer`` whose
``mcts.seed`` is ``rollout_seed(seed, ply) ^ slot`` uses: a batched game equals the single-game route move for move (tests/test_gpu_parity.py::test_batched_evaluation_equals_single_games).
"""
import numpy as np

from .selfplay import _splitmix64, batch_pi_and_moves, move_uniform


def rollout_seed(seed, ply):
    """31-bit seed of the play-out generator for the pure-MCTS moves at ``ply`` (one per ply, like the one integer a single-game
    RolloutPlayer draws per move)."""
    with np.errstate(over='ignore'):
        x = _splitmix64(_splitmix64(np.uint64(seed)) ^ np.uint64(0x726F6C6C00000000 + int(ply)))
    return int(x) & 0x7fffffff


class DuelResult(object):
    """One finished evaluation game: the moves, the winner id (0 = first mover, 1 = second, -1 = tie) and who moved first."""

    def __init__(self, game_id, moves, winner, net_first):
        self.game_id, self.moves, self.winner, self.net_first = int(game_id), [int(m) for m in moves], int(winner), bool(net_first)

    @property
    def net_won(self):
        return self.winner == (0 if self.net_first else 1)


class BatchedEvaluation(object):
    """``n_games`` evaluation games in flight; see the module docstring."""

    def __init__(self, net_engine, net_evaluator, rollout_engine, temperature=1e-3, seed=0, n_playout=None,
                 rollout_playouts=None, n_limit=1000, use_graph=False, sims_per_graph=8):
        a, b = net_engine, rollout_engine
        if (a.game, a.board_size, a.n_in_row, a.n_games) != (b.game, b.board_size, b.n_in_row, b.n_games):
            raise ValueError('the two engines must hold the same games')
        torch = a.torch
        self.torch = torch
        self.net_eng, self.net_evaluator, self.ro_eng = a, net_evaluator, b
        self.net_stream, self.ro_stream = torch.cuda.Stream(device=a.device), torch.cuda.Stream(device=b.device)
        for s in (self.net_stream, self.ro_stream):
            s.wait_stream(torch.cuda.current_stream(a.device))
        self.temperature = float(temperature)
        self.seed = int(seed)
        self.n_playout = int(n_playout if n_playout is not None else a.n_playout)
        self.rollout_playouts = int(rollout_playouts if rollout_playouts is not None else b.n_playout)
        if self.n_playout > a.n_playout or self.rollout_playouts > b.n_playout:
            raise ValueError('an engine is sized for fewer simulations than asked for')
        self.n_limit = int(n_limit)
        self.use_graph, self.sims_per_graph = bool(use_graph), int(sims_per_graph)
        self.n_slots = a.n_games
        if self.use_graph:
            with torch.cuda.stream(self.net_stream):
                a.reset_games()
                a.warm_graph(net_evaluator, a.graph_chunk(self.sims_per_graph))
            torch.cuda.synchronize()

    @classmethod
    def for_network(cls, net_module, board, n_in_row, n_games, n_playout, rollout_playouts=1000, c_puct=5.0,
                    rollout_c_puct=5.0, device='cuda:0', game='gomoku', net_shape=None, temperature=1e-3, seed=0,
                    n_limit=1000, use_graph=True, sims_per_graph=8, **engine_kw):
        """The two engines (+ the hand-written evaluator of ``net_module``, a PolicyValueNet) for ``n_games`` games in flight.
        ``c_puct`` / ``rollout_c_puct``: of the network player and of the pure-MCTS player (tools/train_alphazero.py:141-144: both 5)."""
        from .engine import HipNetEvaluator, MCTSEngine
        net_eng = MCTSEngine(board, n_in_row, n_games=n_games, n_playout=n_playout, c_puct=c_puct, device=str(device),
                             game=game, add_noise=False, **engine_kw)
        ro_eng = MCTSEngine(board, n_in_row, n_games=n_games, n_playout=rollout_playouts, c_puct=rollout_c_puct,
                            device=str(device), game=game, add_noise=False, **engine_kw)
        ev = HipNetEvaluator(net_module, net_shape if net_shape is not None else board, str(device), max_boards=n_games)
        return cls(net_eng, ev, ro_eng, temperature=temperature, seed=seed, n_limit=n_limit, use_graph=use_graph,
                   sims_per_graph=sims_per_graph)

    def refresh_weights(self):
        """Re-upload the network weights if the torch module changed (after a learner step)."""
        refresh = getattr(self.net_evaluator, 'refresh_if_changed', None)
        if refresh is not None:
            refresh(content=True)

    def close(self):
        for eng in (self.net_eng, self.ro_eng):
            eng.close()

    # ------------------------------------------------------------------------------------------------------------------
    def _legal(self, taken):
        eng = self.net_eng
        if eng.game == 'connect4':  # action = column, legal while its top cell is empty
            return ~taken[:, (eng.rows - 1) * eng.cols:]
        return ~taken

    def _cells(self, taken, chosen):
        eng = self.net_eng
        if eng.game != 'connect4':
            return chosen
        heights = taken.reshape(len(chosen), eng.rows, eng.cols).sum(axis=1)  # the stone drops
        return heights[np.arange(len(chosen)), chosen] * eng.cols + chosen

    def run(self, game_ids=None, net_first=None, max_moves=None):
        """Play ``game_ids`` (default 0 .. n_slots-1; at most n_slots; game g sits in slot i of the list) to the end.
        ``net_first``: bool per game, True (default: every game, as the reference seats them) = the network player is player 0.
        -> list of DuelResult in the order of ``game_ids``."""
        torch, G = self.torch, self.n_slots
        ids = list(range(G)) if game_ids is None else [int(g) for g in game_ids]
        n = len(ids)
        if n > G:
            raise ValueError('%d games for %d slots' % (n, G))
        first = np.ones(n, dtype=bool) if net_first is None else np.asarray(net_first, dtype=bool)
        if first.shape != (n,):
            raise ValueError('net_first: one flag per game')
        gid = np.asarray(ids, dtype=np.int64)
        running = np.zeros(G, dtype=bool)
        running[:n] = True
        net_is_0 = np.ones(G, dtype=bool)
        net_is_0[:n] = first
        taken = np.zeros((G, self.net_eng.n_cells), dtype=bool)
        moves_of = [[] for _ in range(n)]
        winner_of = np.full(n, -1, dtype=np.int64)
        for eng, stream in ((self.net_eng, self.net_stream), (self.ro_eng, self.ro_stream)):
            with torch.cuda.stream(stream):
                eng.reset_games()
        ply = 0
        while running.any() and (max_moves is None or ply < max_moves):
            net_turn = running & (net_is_0 == (ply % 2 == 0))
            ro_turn = running & ~net_turn
            # -- both searches, each on its stream over its own games
            if net_turn.any():
                with torch.cuda.stream(self.net_stream):
                    self.net_eng.set_active(net_turn.astype(np.uint8))
                    self.net_eng.simulate(self.net_evaluator, self.n_playout, use_graph=self.use_graph,
                                          sims_per_graph=self.net_eng.graph_chunk(self.sims_per_graph))
            if ro_turn.any():
                from .engine import RolloutEvaluator
                with torch.cuda.stream(self.ro_stream):
                    self.ro_eng.set_active(ro_turn.astype(np.uint8))
                    self.ro_eng.simulate(RolloutEvaluator(rollout_seed(self.seed, ply), self.n_limit), self.rollout_playouts)
            # -- the moves
            chosen = np.full(G, -1, dtype=np.int64)
            if net_turn.any():
                with torch.cuda.stream(self.net_stream):
                    visits = self.net_eng.root_visits()
                    if hasattr(getattr(self.net_evaluator, 'hip', None), 'check_flags'):
                        self.net_evaluator.hip.check_flags()
                rows = np.nonzero(net_turn)[0]
                # the first draw is made and dropped, the second is played (alphazero_mcts.py:148,157); one pi for both
                u_played = move_uniform(self.seed, gid[rows], np.full(len(rows), 2 * ply + 1))
                _, picked = batch_pi_and_moves(visits[rows], self._legal(taken[rows]), self.temperature, u_played)
                chosen[rows] = picked
            if ro_turn.any():
                with torch.cuda.stream(self.ro_stream):
                    visits = self.ro_eng.root_visits()
                rows = np.nonzero(ro_turn)[0]
                chosen[rows] = np.where(self._legal(taken[rows]), visits[rows], -1).argmax(axis=1)  # first maximum
            rows = np.nonzero(running)[0]
            taken[rows, self._cells(taken[rows], chosen[rows])] = True
            for s in rows:
                moves_of[s].append(int(chosen[s]))
            # -- fresh roots for the searched trees (both players reset after every move), the move on both engines' boards
            step_moves = chosen.astype(np.int32)
            verdicts = []
            for eng, stream, turn in ((self.net_eng, self.net_stream, net_turn), (self.ro_eng, self.ro_stream, ro_turn)):
                with torch.cuda.stream(stream):
                    eng.advance(np.where(turn, -1, -2).astype(np.int32))
                    verdicts.append(eng.step(step_moves))
            (winner, ended), (winner_b, ended_b) = verdicts
            if not (np.array_equal(winner[rows], winner_b[rows]) and np.array_equal(ended[rows], ended_b[rows])):
                raise RuntimeError('the two engines disagree about a board')   # (they are stepped with the same moves)
            for s in rows:
                if ended[s]:
                    winner_of[s] = winner[s]
                    running[s] = False
            ply += 1
        self.net_eng.check()
        self.ro_eng.check()
        return [DuelResult(ids[i], moves_of[i], winner_of[i], first[i]) for i in range(n)]
