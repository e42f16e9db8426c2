"""Host driver of the HIP MCTS engine: device buffers, the simulation loop, evaluators.

PyTorch is used for device memory, streams and (for the evaluator) the network forward;
the tree, the board rules and the observation encoding run in the hand-written HIP kernels
of ``csrc/rz_engine.hip`` through the C ABI of ``include/rlzero_hip.h``.

One simulation step of all games = what the reference does once per game in
``AlphaZeroMCTS._playout`` (rlzero/mcts/alphazero_mcts.py:42-71):

    rz_select_step      select loop + env.step + game_end_winner + current_state
    evaluator           policy_value_fn on the batch of leaves
    rz_expand_backup    expand / terminal value + update_recursive
"""
import ctypes
import os

import numpy as np

from . import _hip
from ._hip import HipError, check

WORDS = _hip.BOARD_WORDS


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_DEVICE_INDEX = {}


def _current_stream(torch, device):
    """The current stream of ``device`` as the void* the C ABI takes: torch's raw accessor when it has one (the Stream-object route
    costs ~4 us a call, a dozen calls per move of the one-game API)."""
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is None:
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    idx = _DEVICE_INDEX.get(device)
    if idx is None:
        d = torch.device(device)
        idx = _DEVICE_INDEX[device] = d.index if d.index is not None else torch.cuda.current_device()
    return ctypes.c_void_p(raw(idx))


class SyntheticEvaluator(object):
    """v0 / vlin of SURVEY.md Appendix B, computed on the device from the leaf bitboards."""
    needs_obs = False

    def __init__(self, kind):
        self.kind = {'v0': _hip.EVAL_V0, 'vlin': _hip.EVAL_VLIN}[kind] if isinstance(kind, str) else kind

    def __call__(self, eng):
        check(eng.lib.rz_eval_synthetic(eng.handle, self.kind, _ptr(eng.logp), _ptr(eng.value),
                                        eng.stream()), 'rz_eval_synthetic')
        return eng.logp, eng.value


class RolloutEvaluator(object):
    """Leaf value from a uniformly random play-out on the device (RolloutMCTS._evaluate,
    rlzero/mcts/rollout_mcts.py:49-74); priors are uniform (None -> rz_expand_backup fills 1/k)."""
    needs_obs = False

    def __init__(self, seed=0, n_limit=1000):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.n_limit = int(n_limit)
        self.sim_index = 0

    def __call__(self, eng):
        check(eng.lib.rz_eval_rollout(eng.handle, self.seed, self.sim_index & 0xFFFFFFFF, self.n_limit,
                                      _ptr(eng.value), eng.stream()), 'rz_eval_rollout')
        self.sim_index += 1
        return None, eng.value


class NetEvaluator(object):
    """Batched forward of a policy-value module on the leaf observations
    (AlphaZeroAgent.policy_value_fn, rlzero/games/gomoku/alphazero_agent.py:31-46, for the
    whole batch at once).  ``net(obs[G,4,B,B]) -> (log_probs[G,S], value[G,1])``."""
    needs_obs = True

    def __init__(self, net):
        self.net = net

    def __call__(self, eng):
        import torch
        with torch.no_grad():
            logp, value = self.net(eng.obs)
        return logp.contiguous(), value.reshape(-1).contiguous()


PARAM_ORDER = ('conv1.weight', 'conv1.bias', 'conv2.weight', 'conv2.bias', 'conv3.weight', 'conv3.bias',
               'act_conv1.weight', 'act_conv1.bias', 'act_fc1.weight', 'act_fc1.bias',
               'val_conv1.weight', 'val_conv1.bias', 'val_fc1.weight', 'val_fc1.bias',
               'val_fc2.weight', 'val_fc2.bias')



def compact_grid_board(rows, cols):
    """The boards k_trunk_split has a compact LDS grid for (rz_net.hip: launch_trunk / rz_net_search_resident): N-tiles of
    min(32 // cols, 16) rows, at most two of them, at most 7 columns, tile rows + the halo inside 15 grid rows."""
    if os.environ.get('RZ_NET_COMPACT', '1') == '0' or cols > 7 or cols < 1:
        return False
    tile_rows = min(32 // cols, 16)
    tiles = (rows + tile_rows - 1) // tile_rows
    return tiles <= 2 and tiles * tile_rows + 2 <= 15


class HipNet(object):
    """The hand-written fused fp32-MFMA forward (csrc/rz_net.hip) of a PolicyValueNet."""

    def __init__(self, board_size, device='cuda:0', max_boards=512):
        import torch
        self.lib = _hip.load()
        self.torch = torch
        dev = torch.device(device)
        if dev.type != 'cuda' or not torch.cuda.is_available():
            raise HipError('HipNet needs an MI355X (device=%r); there is no CPU fallback' % (device, ))
        self.device = torch.device('cuda', dev.index if dev.index is not None else torch.cuda.current_device())
        if isinstance(board_size, (tuple, list)):  # (rows, cols, n_actions), e.g. Connect4 (6, 7, 7)
            self.rows, self.cols, self.n_actions = (int(v) for v in board_size)
        else:
            self.rows = self.cols = int(board_size)
            self.n_actions = self.rows * self.cols
        self.board_size, self.n_cells = board_size, self.rows * self.cols
        handle = ctypes.c_void_p()
        check(self.lib.rz_net_create(self.rows, self.cols, self.n_actions, self.device.index,
                                     ctypes.byref(handle)), 'rz_net_create')
        self.handle = handle
        self.max_boards = 0
        self._want = int(max_boards)

    def set_algo(self, algo):
        """conv2 / conv3 algorithm: 'split_f16' (default: direct convolution on the f16 matrix pipe, every f32
        operand carried as a hi + lo pair of f16 values, f32 accumulation -- as accurate as 'direct'), or on the
        f32-input MFMA: 'winograd_f4' (F(4x4,3x3)) or 'direct' (bit-for-bit a k-ordered fmaf chain).  'split_f16_tiles'
        is 'split_f16' with the 32 x 32 x 16 tile kernel on every board size (boards of 11 .. 16 rows and columns otherwise run the row-tile kernel).
        'split_f16_fp8' (OPT-IN, narrower than the reference's f32; boards of 11 .. 16 rows and columns, positions only): 'split_f16'
        with conv3's cross terms hi x lo + lo x hi on the block-scaled FP8 pipe (include/rlzero_hip.h: RZ_NET_SPLIT_F16_FP8)."""
        code = {'direct': _hip.NET_DIRECT, 'winograd_f4': _hip.NET_WINOGRAD_F4, 'split_f16': _hip.NET_SPLIT_F16,
                'split_f16_tiles': _hip.NET_SPLIT_F16_TILES, 'split_f16_fp8': _hip.NET_SPLIT_F16_FP8}[algo]
        check(self.lib.rz_net_set_algo(self.handle, code), 'rz_net_set_algo')
        self.algo = algo
        return self

    def reads_positions(self):
        """True when the trunk can be fed the engine's leaf bitboards (rz_net_trunk_leaves): the 'split_f16' trunk of a
        net with finite activation bounds.  The tree kernels then write no observation planes at all."""
        return getattr(self, 'algo', 'split_f16') in ('split_f16', 'split_f16_tiles', 'split_f16_fp8') and getattr(self, '_split_ok', True)

    def trunk_leaves(self, eng):
        """The trunk on the engine's current leaves, read as bitboards (no float planes), into the internal buffer."""
        n = eng.n_leaves
        if n > self.max_boards:
            self.reserve(n)
        stones, to_move, last = eng.leaf_buffers()
        check(self.lib.rz_net_trunk_leaves(self.handle, stones, to_move, last, n, self._stream()), 'rz_net_trunk_leaves')

    # -- deferred priors (include/rlzero_hip.h: rz_value_head) -------------------------------------------------
    def supports_deferred(self):
        """True when this net's trunk can leave the policy features in a store and hand the tree step the value head's inputs
        (rz_net_trunk_leaves_deferred): the 'split_f16' trunks, every board size."""
        return getattr(self, 'algo', 'split_f16') in ('split_f16', 'split_f16_tiles', 'split_f16_fp8') and getattr(self, '_split_ok', True)

    def deferred_bytes_per_slot(self, n_boards):
        """Device bytes one store slot (one simulation step of ``n_boards`` leaves) takes: f16 feature pieces + logits."""
        tiles = (n_boards + 63) // 64 * 2
        k_steps = (4 * self.n_cells + 15) // 16
        n_pad = (self.n_actions + 31) // 32 * 32
        return tiles * k_steps * 2048 + tiles * 32 * n_pad * 4

    def deferred_reserve(self, n_boards, slots):
        """-> True when the store was (re)allocated (rz_net_deferred_reserve grows it when either number exceeds what it holds): device
        addresses captured in hipGraphs before that are stale."""
        have = getattr(self, '_store', (0, 0))
        check(self.lib.rz_net_deferred_reserve(self.handle, int(n_boards), int(slots)), 'rz_net_deferred_reserve')
        moved = int(n_boards) > have[0] or int(slots) > have[1]
        if moved:
            self._store = (max(have[0], int(n_boards)), max(have[1], int(slots)))
        return moved and have != (0, 0)

    def trunk_leaves_deferred(self, eng):
        """The trunk on the engine's current leaves: policy features into the store slot of each game, the value head's
        inputs for the tree step -> RzValueHead."""
        stones, to_move, last = eng.leaf_buffers()
        out = _hip.RzValueHead()
        check(self.lib.rz_net_trunk_leaves_deferred(self.handle, stones, to_move, last, eng.n_leaves, eng.deferred_slot_ptr(),
                                                    ctypes.byref(out), self._stream()), 'rz_net_trunk_leaves_deferred')
        return out

    # -- receptive-field leaf evaluation (include/rlzero_hip.h: rz_net_delta_*) ------------------------------------
    def supports_delta(self):
        """True when leaves can be evaluated against cached bases of the root (boards of 11 .. 16 rows and columns, 'split_f16')."""
        return (getattr(self, 'algo', 'split_f16') == 'split_f16' and getattr(self, '_split_ok', True)
                and 11 <= self.rows <= 16 and 11 <= self.cols <= 16)

    def delta_reserve(self, n_games):
        check(self.lib.rz_net_delta_reserve(self.handle, int(n_games)), 'rz_net_delta_reserve')
        self._delta_games = max(getattr(self, '_delta_games', 0), int(n_games))

    def delta_invalidate(self):
        check(self.lib.rz_net_delta_invalidate(self.handle, self._stream()), 'rz_net_delta_invalidate')

    def delta_bases(self, stones, to_move, n_games):
        """The two bases of every game from its ROOT position (device pointers: uint64 [n][2][4], int32 [n])."""
        check(self.lib.rz_net_delta_bases(self.handle, stones, to_move, int(n_games), self._stream()), 'rz_net_delta_bases')

    def delta_leaves(self, stones, to_move, last, n, slot_of=None, active=None, feat32=None, without_base=False, want_head=True):
        """The trunk on ``n`` leaves against the cached bases (same bits as trunk_leaves_deferred) -> RzValueHead or None."""
        out = _hip.RzValueHead() if want_head else None
        check(self.lib.rz_net_delta_leaves(self.handle, stones, to_move, last, int(n), slot_of, active, feat32, 1 if without_base else 0,
                                           ctypes.byref(out) if out is not None else None, self._stream()), 'rz_net_delta_leaves')
        return out

    def delta_bases_engine(self, eng):
        """The bases of every game of ``eng`` from its root positions (rz_net_delta_bases_engine)."""
        check(self.lib.rz_net_delta_bases_engine(self.handle, eng.handle, self._stream()), 'rz_net_delta_bases_engine')

    def delta_step(self, eng):
        """trunk_leaves_deferred on the engine's leaves through the receptive-field kernel (same bits) -> RzValueHead."""
        out = _hip.RzValueHead()
        check(self.lib.rz_net_delta_step(self.handle, eng.handle, ctypes.byref(out), self._stream()), 'rz_net_delta_step')
        return out

    def delta_trunk_engine(self, eng):
        """trunk_leaves on the engine's leaves through the receptive-field kernel: the FC GEMM's f16 tiles (heads_gemm next)."""
        check(self.lib.rz_net_delta_trunk_engine(self.handle, eng.handle, self._stream()), 'rz_net_delta_trunk_engine')

    def delta_stats(self, reset=False):
        """{'delta': leaves evaluated against a base, 'no_base': leaves that took the four passes, 'tiles3' / 'tiles2': conv3 / conv2 tiles of
        16 cells, 'cells': changed cells, 'resident_sclk_ghz': the shader clock the last resident search ran at}"""
        out = (ctypes.c_uint32 * 8)()
        check(self.lib.rz_net_delta_stats(self.handle, out, 1 if reset else 0), 'rz_net_delta_stats')
        ghz = (256.0 * out[6]) / (10.0 * out[7]) if out[7] else None   # (cycles per nanosecond of the last resident launch's workgroup 0)
        return {'delta': int(out[0]), 'no_base': int(out[1]), 'tiles3': int(out[2]), 'cells': int(out[3]), 'tiles2': int(out[4]),
                'resident_sclk_ghz': ghz}

    def compact_resident(self):
        """Boards whose resident search runs on the COMPACT LDS grid (k_trunk_split<.., RES, 9, 15>: 69 KB, TWO games per CU and any
        number of games per launch): rz_net_search_resident's rule -- at most two N-tiles, at most 7 columns, tile rows + halo within 15
        (3x3 .. 7x7, Connect4's 6x7); RZ_NET_COMPACT=0 switches the grid off."""
        return compact_grid_board(self.rows, self.cols) and self.supports_resident()

    def supports_resident(self):
        """True when whole searches can run as ONE launch, one workgroup per game (rz_net_search_resident)."""
        if getattr(self, 'algo', 'split_f16') not in ('split_f16', 'split_f16_fp8') or not getattr(self, '_split_ok', True):
            return False
        return (11 <= self.rows <= 16 and 11 <= self.cols <= 16) or (self.rows <= 10 and self.cols <= 10)

    def search_resident(self, eng, n_sims, select_first=False):
        """``n_sims`` simulations of every active game of ``eng`` in one launch; the first leaves come from rz_select_step before
        the call, or (``select_first``) from the launch itself."""
        check(self.lib.rz_net_search_resident(self.handle, eng.handle, int(n_sims), 1 if select_first else 0, self._stream()), 'rz_net_search_resident')

    def deferred_gemm(self, n_boards, n_slots):
        """act_fc1 over the stored leaves of slots [0, n_slots) as one GEMM -> RzDeferredLogits."""
        out = _hip.RzDeferredLogits()
        check(self.lib.rz_net_deferred_gemm(self.handle, int(n_boards), int(n_slots), ctypes.byref(out), self._stream()),
              'rz_net_deferred_gemm')
        return out

    def range_info(self):
        """What rz_net_load derived from the weights for the split-f16 trunk: bounds on the activations of conv1 /
        conv2 / the head features for observation planes in [0, 1], the power-of-two scales their f16 pieces are
        stored with (bound x scale < 60000: no overflow is possible on MCTS leaves, whatever the weights), and
        whether the bounds are finite (otherwise 'split_f16' runs the 'direct' kernel for this net)."""
        out = (ctypes.c_float * 8)()
        check(self.lib.rz_net_range_info(self.handle, out), 'rz_net_range_info')
        return {'bounds': (out[0], out[1], out[2]), 'scales': (out[3], out[4], out[5]), 'split_ok': bool(out[6])}

    def check_flags(self):
        """Raise if a kernel reported a condition since the last call (synchronises).  The split-f16 trunk cannot
        overflow on observation planes in [0, 1] (see range_info); an input beyond that range can."""
        flags = ctypes.c_uint32(0)
        check(self.lib.rz_net_error_flags(self.handle, ctypes.byref(flags)), 'rz_net_error_flags')
        if flags.value & _hip.NET_FLAG_F16_RANGE:
            raise HipError("an input outside [0, 1] drove an activation out of the range of the split-f16 trunk: "
                           "use set_algo('direct') for such inputs")
        return self

    def set_heads_algo(self, algo):
        """GEMM of the first FC layers: 'f32' (f32-input MFMA), 'split32' / 'split64' (f16 matrix pipe with hi + lo
        operand pairs after the 'split_f16' trunk, 32 / 64 boards per workgroup), 'parts' (the same arithmetic as
        single-wave workgroups per K quarter, no LDS: fits beside a resident trunk workgroup of another lane; the consumer
        adds the four partial sums), 'auto' (default after the 'split_f16' trunk: 'split64' beside a capped trunk,
        otherwise 'parts' up to 256 boards and 'split32' above; 'f32' after the f32 trunks; with FC weights of at most 40 KB (6x6, Connect4) an
        un-capped batch of at most one board per CU runs 'in_trunk': every trunk workgroup does these layers on its own board,
        no GEMM launch -- selectable up to 10 rows).  All give the same bits."""
        code = {'auto': _hip.NET_HEADS_AUTO, 'f32': _hip.NET_HEADS_F32, 'split32': _hip.NET_HEADS_SPLIT_32,
                'split64': _hip.NET_HEADS_SPLIT_64, 'parts': _hip.NET_HEADS_SPLIT_PARTS, 'in_trunk': _hip.NET_HEADS_IN_TRUNK}[algo]
        check(self.lib.rz_net_set_heads_algo(self.handle, code), 'rz_net_set_heads_algo')
        self.heads_algo = algo
        return self

    def set_max_workgroups(self, n):
        """Cap the persistent trunk workgroups (0 = one per CU) so that the other CUs stay free for
        the tree / FC kernels of a second lane of games (see BatchedSelfPlay)."""
        check(self.lib.rz_net_set_max_workgroups(self.handle, int(n)), 'rz_net_set_max_workgroups')
        return self

    def load_state_dict(self, state_dict):
        """Upload (and re-pack into MFMA fragment order) the 16 tensors of a
        PolicyValueNet.state_dict(); call again after every optimiser step."""
        arrays = []
        for name in PARAM_ORDER:
            t = state_dict[name]
            a = t.detach().to('cpu', self.torch.float32).contiguous().numpy() if hasattr(t, 'detach') \
                else np.ascontiguousarray(t, dtype=np.float32)
            arrays.append(a)
        ptrs = (ctypes.c_void_p * 16)(*[a.ctypes.data for a in arrays])
        check(self.lib.rz_net_load(self.handle, ptrs, 16), 'rz_net_load')
        self.reserve(max(self._want, self.max_boards))
        self._split_ok = self.range_info()['split_ok']
        return self

    def reserve(self, max_boards):
        check(self.lib.rz_net_reserve(self.handle, int(max_boards)), 'rz_net_reserve')
        self.max_boards = max(self.max_boards, int(max_boards))

    def forward(self, obs, logp=None, value=None):
        """obs float32 [n,4,B,B] on the device -> (log_probs [n,S], value [n])."""
        t = self.torch
        n = obs.shape[0]
        if n > self.max_boards:
            self.reserve(n)
        if logp is None:
            logp = t.empty((n, self.n_actions), dtype=t.float32, device=self.device)
        if value is None:
            value = t.empty(n, dtype=t.float32, device=self.device)
        st = ctypes.c_void_p(t.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.rz_net_forward(self.handle, _ptr(obs), n, _ptr(logp), _ptr(value), st), 'rz_net_forward')
        return logp, value

    def _stream(self):
        return _current_stream(self.torch, self.device)

    def trunk_internal(self, obs):
        """k_trunk alone into the internal feature buffer (pair with ``heads``)."""
        n = obs.shape[0]
        if n > self.max_boards:
            self.reserve(n)
        check(self.lib.rz_net_trunk(self.handle, _ptr(obs), n, None, self._stream()), 'rz_net_trunk')

    def heads_gemm(self, n):
        """Only the FC GEMM on the internal features -> the rz_raw_heads record (device pointers to the raw policy
        logits / value hidden layer, or to their four K-quarter partial sums with 'parts') for the tree kernels that
        finish the heads themselves."""
        out = _hip.RzRawHeads()
        check(self.lib.rz_net_heads_gemm(self.handle, int(n), ctypes.byref(out), self._stream()), 'rz_net_heads_gemm')
        return out

    def heads(self, n, logp, value):
        check(self.lib.rz_net_heads(self.handle, int(n), _ptr(logp), _ptr(value), self._stream()), 'rz_net_heads')
        return logp, value

    def trunk(self, obs):
        t = self.torch
        n = obs.shape[0]
        feat = t.empty((n, 6, self.n_cells), dtype=t.float32, device=self.device)
        st = ctypes.c_void_p(t.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.rz_net_trunk(self.handle, _ptr(obs), n, _ptr(feat), st), 'rz_net_trunk')
        return feat

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.rz_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipNetEvaluator(object):
    """Evaluator running the leaf batch through HipNet (weights taken from a torch
    PolicyValueNet; ``refresh()`` re-uploads them after training).  ``fused_heads``: the engine
    lets the tree kernel finish log_softmax / tanh (one launch less per simulation).  On that route the default
    trunk reads the leaf POSITIONS (the engine's bitboards) and ``needs_obs`` is False: no observation planes are
    written or read; the f32 trunks and the un-fused ``__call__`` route take the float planes."""
    fused_heads = True
    # Deferred priors (RZ_SCORE_UCT_REF, one simulation in flight, boards of 11 .. 16 rows): a simulation step is trunk ->
    # tree step; the policy half of the evaluation and the priors of the expanded nodes are written in one batch before
    # anything reads them (MCTSEngine.flush_deferred).  False: the three-launch route (trunk -> FC GEMM -> tree step).
    deferred_priors = True

    @property
    def needs_obs(self):
        return not (self.use_positions and self.hip.reads_positions())

    def deferred_ok(self, eng):
        """Whether ``eng``'s simulation steps take the deferred-priors route with this evaluator."""
        return (self.deferred_priors and self.use_positions and eng.score_mode == _hip.SCORE_UCT_REF and eng.sims_in_flight == 1
                and self.hip.supports_deferred())

    # Receptive-field leaf evaluation (csrc/rz_delta.h): on boards of 11 .. 16 rows and columns the deferred route's trunk recomputes only
    # the windows around the stones a leaf adds to its search's root, on top of activations of the root cached once per move
    # ("bases") -- the same bits as the full kernel.  RZ_NET_DELTA=0 (or delta_trunk = False): the full kernel on every leaf.
    delta_trunk = True

    def delta_ok(self, eng):
        return (self.delta_trunk and os.environ.get('RZ_NET_DELTA', '1') != '0' and self.hip.supports_delta()
                and eng.rows == self.hip.rows and eng.cols == self.hip.cols)

    def prepare_search(self, eng):
        """Before the steps of a search of ``eng``: the bases of its CURRENT roots (the engine counts what moves them: roots_epoch).
        Cheap when nothing moved.  A missed call costs time only: a leaf whose base is stale takes the kernel's route without one."""
        if not self.delta_ok(eng) or eng.sims_in_flight != 1 or eng._capturing or not (self.deferred_ok(eng) or self.delta_three_launch_ok(eng)):
            return
        key = (id(eng), eng.roots_epoch, id(eng.handle))
        if getattr(self, '_delta_key', None) == key:
            return
        if getattr(self.hip, '_delta_games', 0) < eng.n_games:
            self.hip.torch.cuda.synchronize(self.hip.device)
            self.hip.delta_reserve(eng.n_games)
            eng._drop_graphs('rz_net_delta_reserve moved the base cache')
        self.hip.delta_bases_engine(eng)
        self._delta_key = key

    def deferred_trunk(self, eng):
        if self.delta_ok(eng):
            self.prepare_search(eng)   # (nothing while a graph is captured: the eager warm-up steps before a capture reserve the cache)
            if getattr(self.hip, '_delta_games', 0) >= eng.n_games:
                return self.hip.delta_step(eng)
        return self.hip.trunk_leaves_deferred(eng)

    # The resident search: for a batch of at most one game per CU the simulations of a search run as ONE launch, one workgroup per
    # game (trunk -> value head -> expand / backup -> selection, no kernel boundary; the deferred route's trees, values and priors).
    resident_search = True

    def resident_delta_ok(self, eng):
        """The resident search with the receptive-field trunk (k_delta_res): TWO games per CU.  RZ_NET_DELTA_RESIDENT=0: never."""
        return self.delta_ok(eng) and os.environ.get('RZ_NET_DELTA_RESIDENT', '1') != '0'

    def resident_ok(self, eng):
        n_cus = self.hip.torch.cuda.get_device_properties(self.hip.device).multi_processor_count
        # (k_delta_res takes any number of games: beyond two per CU the launch runs in rounds; plan_lanes says when that pays)
        return (self.resident_search and self.deferred_ok(eng) and self.hip.supports_resident()
                and (self.resident_per_cu(eng) >= 2 or eng.n_games <= n_cus))

    def resident_per_cu(self, eng):
        """Resident workgroups a CU holds: two of k_delta_res (boards of 11 .. 16 rows and columns) and of the compact-grid kernel (boards
        of up to 7 columns), one otherwise.  With two, a launch takes ANY number of games: beyond 2 x CUs it runs in rounds."""
        return 2 if (self.resident_delta_ok(eng) or (self.hip.compact_resident() and eng.rows == self.hip.rows and eng.cols == self.hip.cols)) else 1

    def search_resident(self, eng, n_sims, select_first=False):
        want = self.resident_delta_ok(eng)
        if getattr(self, '_delta_res_set', None) != want:
            check(self.hip.lib.rz_net_delta_resident(self.hip.handle, 1 if want else 0), 'rz_net_delta_resident')
            self._delta_res_set = want
        if want and getattr(self.hip, '_delta_games', 0) < eng.n_games and not eng._capturing:
            self.hip.torch.cuda.synchronize(self.hip.device)
            self.hip.delta_reserve(eng.n_games)   # (the library then runs k_delta_res and builds the roots' bases itself)
            eng._drop_graphs('rz_net_delta_reserve moved the base cache')
        self.hip.search_resident(eng, n_sims, select_first)

    def delta_three_launch_ok(self, eng):
        """The three-launch step (trunk -> FC GEMM -> tree step: the PUCT rule, or deferred_priors = False) with the receptive-field
        trunk: positions, one simulation in flight per tree, the f16 FC GEMM."""
        return (self.delta_ok(eng) and not self.needs_obs and eng.sims_in_flight == 1 and not self.deferred_ok(eng)
                and getattr(self.hip, 'heads_algo', 'auto') != 'f32')

    def raw_heads(self, eng):
        if not self.needs_obs:
            if self.delta_three_launch_ok(eng):
                self.prepare_search(eng)   # (nothing while a graph is captured: the eager steps before a capture reserve the cache)
                if getattr(self.hip, '_delta_games', 0) >= eng.n_games:
                    self.hip.delta_trunk_engine(eng)
                    return self.hip.heads_gemm(eng.n_leaves)
            self.hip.trunk_leaves(eng)
        else:
            self.hip.trunk_internal(eng.obs)
        return self.hip.heads_gemm(eng.n_leaves)

    def __init__(self, net_module, board_size, device='cuda:0', max_boards=512, use_positions=True):
        self.module = net_module
        self.use_positions = bool(use_positions)
        self.hip = HipNet(board_size, device, max_boards)
        self.refresh()

    def _fingerprint(self, content=False):
        # in-place optimiser steps bump Tensor._version; load_state_dict / .to() change data_ptr.  The per-move check walks a cached
        # list of the module's Parameter objects (nn.Module.parameters() is a generator over every submodule: 17 calls per move of
        # the one-game API); the per-round content check below takes the list afresh and drops a cache that no longer matches
        params = getattr(self, '_params', None)
        if params is None or content:
            fresh = list(self.module.parameters())
            if params is None or len(fresh) != len(params) or any(a is not b for a, b in zip(fresh, params)):
                params = self._params = fresh
        marks = tuple((p.data_ptr(), p._version) for p in params)
        if not content:
            return marks
        # a write through ``p.data`` (which carries a version counter of its own) changes neither mark: the CONTENT as two float64
        # reductions over the whole parameter set (0.3 M values and one device round trip: per collection round, not per move)
        t = self.hip.torch
        with t.no_grad():
            flat = t.cat([p.detach().reshape(-1) for p in params]).double()
            ramp = t.arange(1, flat.numel() + 1, dtype=flat.dtype, device=flat.device)
            return marks, tuple(t.stack([flat.sum(), (flat * ramp).sum()]).tolist())

    def refresh(self):
        for eng in list(getattr(self, '_deferring', ())):   # stored features belong to the weights they were computed with
            if eng._def_pending > 0 and eng._def_ev is self:   # (rare: a search still pending when the learner steps)
                self.hip.torch.cuda.synchronize(self.hip.device)
                eng.flush_deferred()
                self.hip.torch.cuda.synchronize(self.hip.device)
        self.hip.load_state_dict(self.module.state_dict())
        self._seen = self._fingerprint()
        self._seen_content = self._fingerprint(content=True)[1]

    def refresh_if_changed(self, content=False):
        """Re-upload the weights if the torch module was trained / reloaded since the last upload.  ``content``: also compare
        the parameter VALUES with those uploaded (a copy through ``p.data`` leaves version and pointer alone)."""
        if self._fingerprint() != self._seen or (content and self._fingerprint(content=True)[1] != self._seen_content):
            self.refresh()

    def __call__(self, eng):
        if not self.needs_obs:  # the select kernels were told to write no planes: encode them for this un-fused call
            check(eng.lib.rz_encode_leaf_obs(eng.handle, _ptr(eng.obs), eng.stream()), 'rz_encode_leaf_obs')
        return self.hip.forward(eng.obs, eng.logp, eng.value)


class HostEvaluator(object):
    """Any ``policy_value_fn(env) -> (iterable[(action, prob)], value)`` callable
    (alphazero_mcts.py:28-31,59), called once per leaf on a materialised env object.
    Slow (one device round trip per simulation) but exact for arbitrary evaluators."""
    needs_obs = False
    returns_probs = True  # the callable's probabilities are stored unchanged (TreeNode.prior)

    def __init__(self, fn, make_env):
        self.fn = fn
        self.make_env = make_env  # (stones0:int, stones1:int, to_move, last_move) -> env
        self.log = None           # optional list of (game, value)

    def __call__(self, eng):
        import torch
        stones, to_move, last, term = eng.get_leaves()
        active = eng.active_host if getattr(eng, '_play_active', None) is None else eng._play_active
        values = np.zeros(eng.n_games, dtype=np.float64)
        probs = np.zeros((eng.n_games, eng.n_actions), dtype=np.float32)
        for g in range(eng.n_games):
            if not active[g]:
                continue
            env = self.make_env(bits_to_int(stones[g, 0]), bits_to_int(stones[g, 1]),
                                int(to_move[g]), int(last[g]))
            priors, value = self.fn(env)  # called on terminal leaves too (:59)
            values[g] = value
            for a, p in priors:
                probs[g, int(a)] = p
            if self.log is not None:
                self.log.append((g, float(value)))
        eng.value64.copy_(torch.from_numpy(values))
        eng.logp.copy_(torch.from_numpy(probs))
        return eng.logp, eng.value64


def bits_to_int(words):
    out = 0
    for j, w in enumerate(words):
        out |= int(w) << (64 * j)
    return out


def int_to_bits(x):
    return [(x >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(WORDS)]


class MCTSEngine(object):
    """``n_games`` independent trees searched in lock-step on one GPU."""

    def __init__(self, board_size, n_in_row, n_games=1, n_playout=1000, c_puct=5.0,
                 device='cuda:0', pool_factor=2.0, score_mode='uct_ref', add_noise=False, noise_seed=0,
                 game='gomoku', sims_in_flight=1, in_flight_impl='level_sync'):
        """``board_size``: int B (Gomoku / TicTacToe, B x B) or (rows, cols) for ``game='connect4'``.

        ``sims_in_flight`` = K > 1 (opt-in, NOT the reference's algorithm, never used by a parity test): K
        simulations of every tree share one evaluator batch of ``n_games * K`` leaves; the nodes of a selected path
        carry a virtual loss (N += 1, W -= 1) until their backup.  The reference runs its simulations strictly one
        after the other (alphazero_mcts.py:82-85): results differ from it as soon as K > 1.  Device evaluators
        only.  ``in_flight_impl``: 'level_sync' (the production kernel: a workgroup of K waves per game) or 'sequential'
        (its one-wave restatement, slower; the two build the same trees bit for bit)."""
        import torch
        self.lib = _hip.load()
        self.torch = torch
        dev = torch.device(device)
        if dev.type != 'cuda' or not torch.cuda.is_available():
            raise HipError('the MCTS engine needs an MI355X (device=%r, cuda available=%s); '
                           'there is no CPU fallback' % (device, torch.cuda.is_available()))
        self.device = torch.device('cuda', dev.index if dev.index is not None else torch.cuda.current_device())
        self.game = game
        if game == 'connect4':
            rows, cols = (6, 7) if board_size is None else (int(board_size[0]), int(board_size[1]))
            self.board_size = (rows, cols)
        else:
            rows = cols = int(board_size)
            self.board_size = rows
        self.rows, self.cols, self.n_in_row = rows, cols, int(n_in_row)
        self.n_cells = rows * cols
        self.n_actions = cols if game == 'connect4' else self.n_cells
        self.n_games, self.n_playout, self.c_puct = int(n_games), int(n_playout), float(c_puct)
        self.score_mode = {'uct_ref': _hip.SCORE_UCT_REF, 'puct': _hip.SCORE_PUCT}[score_mode] \
            if isinstance(score_mode, str) else int(score_mode)
        self.add_noise = bool(add_noise)
        self.sims_in_flight = max(1, int(sims_in_flight))
        self.n_leaves = self.n_games * self.sims_in_flight  # rows of obs / logp / value
        cfg = _hip.RzConfig(abi_version=_hip.ABI_VERSION,
                            game_kind=_hip.GAME_CONNECT4 if game == 'connect4' else _hip.GAME_GOMOKU,
                            board_size=rows if game != 'connect4' else 0,
                            n_in_row=self.n_in_row, n_games=self.n_games, n_playout=self.n_playout,
                            score_mode=self.score_mode, add_noise=1 if add_noise else 0, c_puct=self.c_puct,
                            pool_factor=float(pool_factor), device=self.device.index,
                            noise_seed=int(noise_seed) & 0x7FFFFFFF, board_height=rows, board_width=cols,
                            sims_in_flight=self.sims_in_flight,
                            in_flight_impl={'level_sync': 0, 'sequential': 1}[in_flight_impl])
        handle = ctypes.c_void_p()
        check(self.lib.rz_create(ctypes.byref(cfg), ctypes.byref(handle)), 'rz_create')
        self.handle = handle
        G, S, GL = self.n_games, self.n_actions, self.n_leaves
        kw = dict(device=self.device)
        self.obs = torch.zeros((GL, 4, rows, cols), dtype=torch.float32, **kw)
        self.logp = torch.zeros((GL, S), dtype=torch.float32, **kw)
        self.value = torch.zeros(GL, dtype=torch.float32, **kw)
        self.value64 = torch.zeros(GL, dtype=torch.float64, **kw)
        self.visits = torch.zeros((G, S), dtype=torch.int32, **kw)
        self.wsum = torch.zeros((G, S), dtype=torch.float64, **kw)
        self.priors = torch.zeros((G, S), dtype=torch.float32, **kw)
        self.root_n = torch.zeros(G, dtype=torch.int32, **kw)
        self.root_w = torch.zeros(G, dtype=torch.float64, **kw)
        self.moves = torch.zeros(G, dtype=torch.int32, **kw)
        self.winner = torch.zeros(G, dtype=torch.int32, **kw)
        self.ended = torch.zeros(G, dtype=torch.uint8, **kw)
        self.stones = torch.zeros((G, 2, WORDS), dtype=torch.int64, **kw)
        self.to_move = torch.zeros(G, dtype=torch.int32, **kw)
        self.last_move = torch.zeros(G, dtype=torch.int32, **kw)
        self.terminal = torch.zeros(G, dtype=torch.int32, **kw)
        self.mask = torch.zeros(G, dtype=torch.uint8, **kw)
        self.active_dev = torch.ones(G, dtype=torch.uint8, **kw)
        self.noise_keys = torch.zeros(G, dtype=torch.int64, **kw)
        self.noise_mask = torch.zeros(G, dtype=torch.uint8, **kw)
        self.active_host = np.ones(G, dtype=np.uint8)
        self._graphs = {}
        self.roots_epoch = 0   # bumped by whatever moves a root position (set_roots, step, the move step on the device): evaluators that cache
                               # per-root work (HipNetEvaluator's receptive-field bases) compare it with the epoch they computed for
        # deferred priors: the evaluator whose store holds this engine's pending leaves, steps since the last flush, capacity
        self._def_ev, self._def_pending, self._def_slots, self._def_slot_ptr, self._capturing = None, 0, 0, None, False
        # cap of an evaluator's store + logits (a flush every slots steps when n_playout needs more): a whole 800-simulation search of
        # 4096 games is 15 GB of 288 -- below the cap the search is one launch and the move one hipGraph (19 MB per slot at 4096 games)
        self.deferred_max_bytes = 32 << 30

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        return _current_stream(self.torch, self.device)

    def close(self):
        if getattr(self, 'handle', None):
            self._graphs = {}
            self.lib.rz_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        out = _hip.RzStats()
        check(self.lib.rz_get_stats(self.handle, ctypes.byref(out)), 'rz_get_stats')
        return out

    def check(self):
        """Raise if any game overflowed its arena / received an illegal move.  Returns the statistics;
        ``reuse_dropped`` counts kept subtrees that exceeded the carry limit (pool_factor * n_playout expanded
        nodes) and were replaced by a fresh root -- a counted deviation from the reference's unbounded tree, not
        an error."""
        st = self.stats()
        if st.error_flags & ~_hip.FLAG_REUSE_DROPPED:
            names = [n for bit, n in _hip.FLAG_NAMES.items() if st.error_flags & bit]
            raise HipError('engine error flags 0x%x (%s), first bad game %d, arena %d/%d slots' %
                           (st.error_flags, ', '.join(names), st.first_bad_game,
                            st.max_slots_used, st.arena_slots))
        return st

    def poll_errors(self):
        """check() for every move of a one-game player: waits for the current stream, reads the OR of the games' flags (8 bytes) and
        only asks for the full statistics -- which name the game and raise -- when a bit is set.  -> subtrees dropped so far."""
        flags, drops = ctypes.c_int32(0), ctypes.c_int32(0)
        check(self.lib.rz_poll_errors(self.handle, ctypes.byref(flags), ctypes.byref(drops), self.stream()), 'rz_poll_errors')
        if flags.value & ~_hip.FLAG_REUSE_DROPPED:
            self.check()
        return drops.value

    # ------------------------------------------------------------------ roots
    def set_roots(self, stones, to_move, last_move, mask=None, reset_trees=False):
        """stones: uint64 [G,2,WORDS]; to_move / last_move: int [G]; mask: bool [G] or None."""
        t = self.torch
        self.flush_deferred()   # (pending priors belong to the trees as they are)
        G, nb = self.n_games, 2 * WORDS * 8
        if getattr(self, '_roots_io', None) is None:
            # boards, sides to move and last cells go up in ONE copy from pinned memory: [G][2][WORDS] int64 | [G] int32 | [G] int32
            host = t.empty(G * (nb + 8), dtype=t.uint8, pin_memory=True)
            dev = t.empty(G * (nb + 8), dtype=t.uint8, device=self.device)
            hn = host.numpy()
            self._roots_io = (host, dev, hn[:G * nb].view(np.uint64).reshape(G, 2, WORDS), hn[G * nb:G * nb + 4 * G].view(np.int32),
                              hn[G * nb + 4 * G:].view(np.int32), t.cuda.Event())
        host, dev, h_stones, h_to_move, h_last, done = self._roots_io
        done.synchronize()   # (the previous upload has left the pinned buffer)
        h_stones[...] = np.asarray(stones, dtype=np.uint64).reshape(G, 2, WORDS)
        h_to_move[:] = to_move
        h_last[:] = last_move
        dev.copy_(host, non_blocking=True)
        done.record(t.cuda.current_stream(self.device))
        mptr = None
        if mask is not None:
            self.mask.copy_(t.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)))
            mptr = _ptr(self.mask)
        base = dev.data_ptr()
        check(self.lib.rz_set_roots(self.handle, ctypes.c_void_p(base), ctypes.c_void_p(base + G * nb), ctypes.c_void_p(base + G * nb + 4 * G),
                                    mptr, 1 if reset_trees else 0, self.stream()), 'rz_set_roots')
        self.roots_epoch += 1

    def reset_games(self, mask=None):
        """Empty boards, player 0 to move, fresh trees for the selected games."""
        G = self.n_games
        self.set_roots(np.zeros((G, 2, WORDS), np.uint64), np.zeros(G, np.int32),
                       np.full(G, -1, np.int32), mask=mask, reset_trees=True)

    def get_roots(self):
        check(self.lib.rz_get_roots(self.handle, _ptr(self.stones), _ptr(self.to_move),
                                    _ptr(self.last_move), self.stream()), 'rz_get_roots')
        return (self.stones.cpu().numpy().view(np.uint64), self.to_move.cpu().numpy(),
                self.last_move.cpu().numpy())

    def set_active(self, active):
        self._play_active = None   # (the host owns the flags again: HostEvaluator reads active_host)
        self.active_host = np.ascontiguousarray(active, dtype=np.uint8).copy()
        self.active_dev.copy_(self.torch.from_numpy(self.active_host))
        check(self.lib.rz_set_active(self.handle, _ptr(self.active_dev), self.stream()), 'rz_set_active')

    def set_noise_keys(self, keys=None, mask=None):
        """The Dirichlet stream of every selected game: ``keys`` uint64 [G] (None = the engine's per-slot default), ``mask`` bool [G] or
        None (all); the games' expansion counters restart.  BatchedSelfPlay keys a game's noise by (seed, game id), so its search under
        the PUCT rule does not depend on the slot, lane or GPU it is played on (rz_set_noise_keys)."""
        t = self.torch
        kptr = mptr = None
        if keys is not None:
            self.noise_keys.copy_(t.from_numpy(np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64)))
            kptr = _ptr(self.noise_keys)
        if mask is not None:
            self.noise_mask.copy_(t.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)))
            mptr = _ptr(self.noise_mask)
        check(self.lib.rz_set_noise_keys(self.handle, kptr, mptr, self.stream()), 'rz_set_noise_keys')

    def get_leaves(self):
        check(self.lib.rz_get_leaves(self.handle, _ptr(self.stones), _ptr(self.to_move),
                                     _ptr(self.last_move), _ptr(self.terminal), self.stream()),
              'rz_get_leaves')
        return (self.stones.cpu().numpy().view(np.uint64), self.to_move.cpu().numpy(),
                self.last_move.cpu().numpy(), self.terminal.cpu().numpy())

    def leaf_buffers(self):
        """Device pointers (stones, to_move, last cell) of the engine's leaf arrays, [n_leaves] rows."""
        if getattr(self, '_leaf_ptrs', None) is None:
            a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            check(self.lib.rz_leaf_buffers(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), 'rz_leaf_buffers')
            self._leaf_ptrs = (a, b, c)
        return self._leaf_ptrs

    def root_obs(self):
        check(self.lib.rz_encode_root_obs(self.handle, _ptr(self.obs), self.stream()), 'rz_encode_root_obs')
        return self.obs

    # ------------------------------------------------------------------ the hot loop
    def sim_step(self, evaluator):
        """One simulation for every active game (enqueued, not synchronised)."""
        ok = getattr(evaluator, 'deferred_ok', None)
        if self.sims_in_flight > 1 or (ok is not None and ok(self)):
            return self.sim_chunk(evaluator, 1)   # (the deferred route's store bookkeeping lives there)
        self.flush_deferred()
        obs = _ptr(self.obs) if getattr(evaluator, 'needs_obs', True) else None
        check(self.lib.rz_select_step(self.handle, obs, self.stream()), 'rz_select_step')
        logp, value = evaluator(self)
        if getattr(evaluator, 'returns_probs', False):
            check(self.lib.rz_expand_backup_probs(self.handle, _ptr(logp), _ptr(value), self.stream()),
                  'rz_expand_backup_probs')
        elif value.dtype == self.torch.float64:
            check(self.lib.rz_expand_backup_f64(self.handle, _ptr(logp), _ptr(value), self.stream()),
                  'rz_expand_backup_f64')
        else:
            check(self.lib.rz_expand_backup(self.handle, _ptr(logp), _ptr(value), self.stream()),
                  'rz_expand_backup')

    def _in_flight(self, k_backup, k_select):
        check(self.lib.rz_set_in_flight(self.handle, int(k_backup), int(k_select)), 'rz_set_in_flight')

    def sim_chunk(self, evaluator, n):
        """``n`` simulations of every active game: select, (evaluate, expand+backup+select) x
        (n-1), evaluate, expand+backup -- consecutive simulations share one tree launch.
        With ``sims_in_flight`` = K > 1: ceil(n / K) steps of K simulations each (the last one with the remainder)."""
        if n <= 0:
            return
        K = self.sims_in_flight
        ok = getattr(evaluator, 'deferred_ok', None)
        deferred = ok is not None and ok(self)
        if not deferred:
            self.flush_deferred()   # (an evaluator of another route takes over: its expansions write their priors at once)
        if isinstance(evaluator, HostEvaluator):
            if K > 1:
                raise HipError('host evaluators are not available with sims_in_flight > 1')
            if getattr(self, 'play_log', None) is not None and getattr(self, '_play_on', False):
                # the move step on the device owns the games' active flags: which slots hold a running game, as of this search
                self._play_active = (self.play_state()[2] == 1).astype(np.uint8)
            for _ in range(n):
                self.sim_step(evaluator)
            return
        lib, h = self.lib, self.handle
        begin = getattr(evaluator, 'begin_chunk', None)  # optional evaluator hook (bench.py's timing wrapper)
        if begin is not None:
            begin()
        if deferred:
            return self._sim_chunk_deferred(evaluator, n)
        obs = _ptr(self.obs) if getattr(evaluator, 'needs_obs', True) else None
        steps = -(-n // K)
        counts = [min(K, n - i * K) for i in range(steps)] + [0]  # slots in flight in step i
        if K > 1:
            self._in_flight(0, counts[0])
        check(lib.rz_select_step(h, obs, self.stream()), 'rz_select_step')
        fused = getattr(evaluator, 'fused_heads', False)
        for i in range(steps):
            if K > 1 and (i == 0 or counts[i + 1] != K or counts[i] != K):
                self._in_flight(counts[i], counts[i + 1])
            if fused:
                heads = evaluator.raw_heads(self)
                if i + 1 < steps:
                    check(lib.rz_tree_step_raw(h, ctypes.byref(heads), obs, self.stream()), 'rz_tree_step_raw')
                else:
                    check(lib.rz_expand_backup_raw(h, ctypes.byref(heads), self.stream()), 'rz_expand_backup_raw')
                continue
            logp, value = evaluator(self)
            if value.dtype != self.torch.float32:
                raise HipError('device evaluators must return float32 values')
            if i + 1 < steps:
                check(lib.rz_tree_step(h, _ptr(logp), _ptr(value), obs, self.stream()), 'rz_tree_step')
            else:
                check(lib.rz_expand_backup(h, _ptr(logp), _ptr(value), self.stream()), 'rz_expand_backup')
        if K > 1:
            self._in_flight(K, K)

    # ------------------------------------------------------------------ deferred priors
    def deferred_slot_ptr(self):
        if self._def_slot_ptr is None:
            p = ctypes.c_void_p()
            check(self.lib.rz_deferred_slots(self.handle, ctypes.byref(p)), 'rz_deferred_slots')
            self._def_slot_ptr = p
        return self._def_slot_ptr

    def _deferred_begin(self, evaluator, n):
        """Before ``n`` more steps on the deferred route: the store of ``evaluator`` holds this engine's pending leaves (another
        evaluator's are flushed first), reserved once for min(n_playout, what deferred_max_bytes allows) steps; a flush when the
        coming steps would not fit.  -> steps that may be enqueued now (<= n)."""
        # an evaluator's store holds the pending leaves of ONE engine at a time (slot = steps since that engine's flush): another
        # engine that searches with the same evaluator has its priors written first
        owner = getattr(evaluator, '_def_owner', None)
        owner = owner() if owner is not None else None
        if owner is not None and owner is not self:
            # (the other engine's steps may be on another stream than the one current here: its flush waits for them)
            self.torch.cuda.synchronize(self.device)
            owner.flush_deferred()
            self.torch.cuda.synchronize(self.device)
        if owner is not self:
            import weakref
            evaluator._def_owner = weakref.ref(self)
        if self._def_ev is not evaluator:
            self.flush_deferred()
            hip = evaluator.hip
            per_slot = hip.deferred_bytes_per_slot(self.n_leaves) + 80 * self.n_games
            slots = int(max(16, min(max(self.n_playout, 16), self.deferred_max_bytes // per_slot)))
            if self._capturing:
                raise HipError('the deferred-priors store must be reserved before a hipGraph capture (warm_graph does that)')
            if slots > self._def_slots:
                check(self.lib.rz_deferred_reserve(self.handle, slots), 'rz_deferred_reserve')
                self._def_slots = slots
                self._def_slot_ptr = None
                self._drop_graphs('rz_deferred_reserve moved the pending records')
            if hip.deferred_reserve(self.n_leaves, self._def_slots):
                # the evaluator's store moved (a larger engine took it over): every engine's captured launches of this evaluator
                # hold the old addresses
                for other in list(getattr(evaluator, '_deferring', ())) + [self]:
                    other._drop_graphs('rz_net_deferred_reserve moved the feature store')
            self._def_ev = evaluator
            refs = getattr(evaluator, '_deferring', None)
            if refs is None:
                import weakref
                refs = evaluator._deferring = weakref.WeakSet()
            refs.add(self)
        if self._capturing:
            return n
        if self._def_pending + min(n, self._def_slots) > self._def_slots:
            self.flush_deferred()
        return min(n, self._def_slots - self._def_pending)

    def _drop_graphs(self, why):
        """Captured launches hold device addresses by value: when a reservation moves a buffer they replay into freed memory.  The
        graphs are dropped; simulate(use_graph=True) then says to call warm_graph again."""
        if self._graphs or getattr(self, '_move_graph', None) is not None:
            self._graphs = {}
            self._move_graph = None   # (warm_move_graph's whole-move graph bakes the same addresses in)
            self._graphs_dropped = why

    def _sim_chunk_deferred(self, evaluator, n):
        """sim_chunk on the deferred-priors route: per step the trunk (policy features into the step's store slot, value inputs
        on) and ONE tree launch (value head, backup, next selection)."""
        lib, h = self.lib, self.handle
        res_ok = getattr(evaluator, 'resident_ok', None)
        resident = res_ok is not None and res_ok(self)
        while n > 0:
            m = self._deferred_begin(evaluator, n)
            if resident:   # the m simulations in ONE launch, one workgroup per game, the first selection included
                evaluator.search_resident(self, m, True)
                if not self._capturing:
                    self._def_pending += m
                    self._def_stream = self.torch.cuda.current_stream(self.device)
                n -= m
                continue
            check(lib.rz_select_step(h, None, self.stream()), 'rz_select_step')
            for i in range(m):
                head = evaluator.deferred_trunk(self)
                if i + 1 < m:
                    check(lib.rz_tree_step_deferred(h, ctypes.byref(head), self.stream()), 'rz_tree_step_deferred')
                else:
                    check(lib.rz_expand_backup_deferred(h, ctypes.byref(head), self.stream()), 'rz_expand_backup_deferred')
            if not self._capturing:
                self._def_pending += m
                self._def_stream = self.torch.cuda.current_stream(self.device)
            n -= m

    def flush_deferred(self):
        """Write the priors of every expansion since the last flush (one GEMM over the stored leaves + one kernel).  Called by
        whatever reads priors or moves trees (advance, set_roots, root_priors, arena) and when the store is full."""
        if self._def_pending <= 0 or self._def_ev is None:
            return
        # the pending steps were enqueued on _def_stream; a caller on another stream (a direct engine user, a read-out from the
        # default stream while a lane searches) gets the flush ordered behind them, and their stream behind the flush
        cur, then = self.torch.cuda.current_stream(self.device), getattr(self, '_def_stream', None)
        if then is not None and then != cur:
            cur.wait_stream(then)
        logits = self._def_ev.hip.deferred_gemm(self.n_leaves, self._def_pending)
        check(self.lib.rz_deferred_flush(self.handle, ctypes.byref(logits), self._def_pending, self.stream()), 'rz_deferred_flush')
        if then is not None and then != cur:
            then.wait_stream(cur)
        self._def_pending = 0

    def _whole_steps(self, n_sims):
        """``n_sims`` rounded down to whole steps of K simulations (at least one step)."""
        K = self.sims_in_flight
        return max(K, int(n_sims) - int(n_sims) % K)

    def graph_chunk(self, sims_per_graph):
        """Simulations a captured chunk holds.  One simulation in flight: ``sims_per_graph``.  K in flight: whole
        steps of K simulations -- as many as ``sims_per_graph`` STEPS, lowered to a divisor of the full steps of a
        search (n_playout // K) so that only the ragged last step is left to eager launches."""
        per = max(1, int(sims_per_graph))
        K = self.sims_in_flight
        if K == 1:
            return per
        full = max(1, self.n_playout // K)
        steps = max(s for s in range(1, min(per, full) + 1) if full % s == 0)
        return K * steps

    def simulate(self, evaluator, n_sims=None, use_graph=False, sims_per_graph=8):
        """Run ``n_sims`` (default n_playout) simulations in every active game.

        use_graph: replay a hipGraph holding ``sims_per_graph`` simulations (captured by
        ``warm_graph``) instead of launching kernel by kernel; device-side evaluators only."""
        n = self.n_playout if n_sims is None else int(n_sims)
        prep = getattr(evaluator, 'prepare_search', None) or getattr(getattr(evaluator, 'inner', None), 'prepare_search', None)
        if prep is not None:
            prep(self)   # (per-root work of the evaluator: a graph replay calls nothing of it)
        res_ok = getattr(evaluator, 'resident_ok', None)
        if not use_graph or isinstance(evaluator, HostEvaluator) or (res_ok is not None and res_ok(self)):
            self.sim_chunk(evaluator, n)   # (the resident search is two launches for the whole search: nothing to capture)
            return
        per = self._whole_steps(sims_per_graph)
        if per > n:
            self.sim_chunk(evaluator, n)
            return
        key = (id(evaluator), per)
        if key not in self._graphs:
            why = getattr(self, '_graphs_dropped', None)
            raise HipError('call warm_graph(evaluator, %d) before simulate(use_graph=True)%s' % (per, ' (the graphs were dropped: %s)' % why if why else ''))
        graph = self._graphs[key][0]
        full, rest = divmod(n, per)
        ok = getattr(evaluator, 'deferred_ok', None)
        deferred = ok is not None and ok(self)
        for _ in range(full):
            if deferred:   # the replay writes `per` more slots of the store
                if self._deferred_begin(evaluator, per) < per:
                    raise HipError('the deferred-priors store (%d slots) is smaller than a graph of %d steps' % (self._def_slots, per))
                self._def_pending += per
                self._def_stream = self.torch.cuda.current_stream(self.device)
            graph.replay()
        self.sim_chunk(evaluator, rest)

    def warm_graph(self, evaluator, per):
        """Capture ``per`` simulations into a hipGraph (torch.cuda.CUDAGraph around our launches
        and the evaluator).  Capturing executes nothing, but the eager warm-up does run 3
        simulations, so call this on throw-away tree state (before reset_games)."""
        t = self.torch
        per = self._whole_steps(per)
        key = (id(evaluator), per)
        if key in self._graphs:
            return self._graphs[key][0]
        res_ok = getattr(evaluator, 'resident_ok', None)
        if res_ok is not None and res_ok(self):
            return None   # (the resident search is one launch per search: simulate() replays no graph for it)
        cur = t.cuda.current_stream(self.device)
        side = t.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with t.cuda.stream(side):
            self.sim_chunk(evaluator, 3 * self.sims_in_flight)
        cur.wait_stream(side)
        t.cuda.synchronize(self.device)
        self.flush_deferred()   # (the warm-up's leaves; a capture holds no flush: simulate() places them between replays)
        graph = t.cuda.CUDAGraph()
        self._capturing = True
        try:
            with t.cuda.graph(graph):
                self.sim_chunk(evaluator, per)
        finally:
            self._capturing = False
        self._graphs[key] = (graph, evaluator)
        return graph

    # ------------------------------------------------------------------ read-out
    def root_visits(self):
        check(self.lib.rz_root_visits(self.handle, _ptr(self.visits), self.stream()), 'rz_root_visits')
        if self._def_pending > 0:
            # the visit counts do not wait for the priors: their copy is enqueued first, the flush runs while the host works on them
            t = self.torch
            if getattr(self, '_visits_host', None) is None:
                self._visits_host = t.empty(self.visits.shape, dtype=self.visits.dtype, pin_memory=True)
            self._visits_host.copy_(self.visits, non_blocking=True)
            done = t.cuda.Event()
            done.record(t.cuda.current_stream(self.device))
            self.flush_deferred()
            done.synchronize()
            return self._visits_host.numpy().copy()
        return self.visits.cpu().numpy()

    def root_wsum(self):
        check(self.lib.rz_root_wsum(self.handle, _ptr(self.wsum), self.stream()), 'rz_root_wsum')
        return self.wsum.cpu().numpy()

    def root_priors(self):
        self.flush_deferred()
        check(self.lib.rz_root_priors(self.handle, _ptr(self.priors), self.stream()), 'rz_root_priors')
        return self.priors.cpu().numpy()

    def root_stats(self):
        check(self.lib.rz_root_stats(self.handle, _ptr(self.root_n), _ptr(self.root_w), self.stream()),
              'rz_root_stats')
        return self.root_n.cpu().numpy(), self.root_w.cpu().numpy()

    def advance(self, moves):
        """update_with_move for every game: move >= 0 keep that subtree, -1 reset, -2 skip."""
        self.flush_deferred()   # (the kept subtree's prior blocks are copied: they must be written)
        t = self.torch
        if getattr(self, '_adv_io', None) is None:
            self._adv_io = (t.empty(self.n_games, dtype=t.int32, pin_memory=True), t.cuda.Event())
        host, done = self._adv_io
        done.synchronize()
        host.numpy()[:] = moves
        self.moves.copy_(host, non_blocking=True)
        done.record(t.cuda.current_stream(self.device))
        check(self.lib.rz_advance_roots(self.handle, _ptr(self.moves), self.stream()), 'rz_advance_roots')

    def advance_and_step(self, keep_moves, step_moves):
        """``advance(keep_moves)`` then ``step(step_moves)`` as one host step: both move vectors go up in one copy from pinned
        memory, winners and ended flags come back in one, and the host waits once (the self-play loop's move: update_with_move
        before the boards change, alphazero_mcts.py:96-103, then env.step + game_end_winner).  -> (winner[G], ended[G])."""
        t = self.torch
        self.flush_deferred()   # (the kept subtree's prior blocks are copied: they must be written)
        G = self.n_games
        if getattr(self, '_move_io', None) is None:
            self._move_io = (t.empty((2, G), dtype=t.int32, pin_memory=True), t.empty((2, G), dtype=t.int32, device=self.device),
                             t.empty(G, dtype=t.int32, pin_memory=True), t.empty(G, dtype=t.uint8, pin_memory=True), t.cuda.Event())
        h_in, d_in, h_win, h_end, done = self._move_io
        h_in[0].numpy()[:] = keep_moves
        h_in[1].numpy()[:] = step_moves
        d_in.copy_(h_in, non_blocking=True)
        check(self.lib.rz_advance_roots(self.handle, _ptr(d_in[0]), self.stream()), 'rz_advance_roots')
        check(self.lib.rz_step_games(self.handle, _ptr(d_in[1]), _ptr(self.winner), _ptr(self.ended), self.stream()), 'rz_step_games')
        self.roots_epoch += 1
        h_win.copy_(self.winner, non_blocking=True)
        h_end.copy_(self.ended, non_blocking=True)
        done.record(t.cuda.current_stream(self.device))
        done.synchronize()
        return h_win.numpy().copy(), h_end.numpy().copy()

    def step(self, moves):
        """env.step + game_end_winner on the root boards -> (winner[G], ended[G])."""
        self.moves.copy_(self.torch.from_numpy(np.ascontiguousarray(moves, dtype=np.int32)))
        check(self.lib.rz_step_games(self.handle, _ptr(self.moves), _ptr(self.winner),
                                     _ptr(self.ended), self.stream()), 'rz_step_games')
        self.roots_epoch += 1
        return self.winner.cpu().numpy(), self.ended.cpu().numpy()

    # ------------------------------------------------------------------ the move step on the device
    def play_attach(self, seed, temperature, queue_ids, queue_ctl, ring_steps=64, stall_margin=0.0):
        """Hand the move step to the device (include/rlzero_hip.h: rz_play_*): ``queue_ids`` int64 / ``queue_ctl`` int32 [2] device
        tensors (the queue of game ids, shared by the lanes of a GPU).  Allocates this engine's log ring -> the tensor
        [ring_steps][n_games][8 + A] int32.  Every slot starts idle; ``play_apply()`` fills them from the queue."""
        t = self.torch
        self.flush_deferred()
        t.cuda.synchronize(self.device)
        # The log ring lives in PINNED HOST memory, which the kernels address directly: a row is written by the move's own kernels
        # (write-only but for the flag of a game that ends) and read by the host behind an event -- no copy command between two
        # moves (an 18-us gap + a blit kernel + 6 us on a lane's stream per read-back).  RZ_PLAY_DEVICE_LOG=1: a device ring that the
        # caller copies from (the first form; kept for comparison).
        shape = (int(ring_steps), self.n_games, _hip.PLAY_RECORD_WORDS + self.n_actions)
        self.play_log_on_host = os.environ.get('RZ_PLAY_DEVICE_LOG') != '1'
        if self.play_log_on_host:
            self.play_log = t.zeros(shape, dtype=t.int32).pin_memory()
        else:
            self.play_log = t.zeros(shape, dtype=t.int32, device=self.device)
        self._play_queue = (queue_ids, queue_ctl)   # (kept alive: the engine holds their addresses)
        def attach():
            cfg = _hip.RzPlayConfig(seed=int(seed) & 0xFFFFFFFFFFFFFFFF, temperature=float(temperature), stall_margin=float(stall_margin),
                                    d_queue_ids=queue_ids.data_ptr(), d_queue_ctl=queue_ctl.data_ptr(), d_log=self.play_log.data_ptr(),
                                    ring_steps=int(ring_steps), reserved=0)
            check(self.lib.rz_play_attach(self.handle, ctypes.byref(cfg)), 'rz_play_attach')
        try:
            attach()
        except HipError as exc:
            if not self.play_log_on_host or 'cannot address' not in str(exc):
                raise
            # (pinned memory this device cannot address -- the library asked the runtime: the device ring and its copies instead)
            self.play_log_on_host = False
            self.play_log = t.zeros(shape, dtype=t.int32, device=self.device)
            attach()
        self.play_steps = 0
        self._play_on, self._play_active = True, None
        self.active_host[:] = 0
        return self.play_log

    def play_move(self):
        """The move step behind a search, enqueued on the current stream without a host round trip: the draw (rz_play_draw: visits
        into the log, the move of alphazero_mcts.py:147-148 or a stall), the priors of the search's expansions (they must be written
        before the kept subtree is copied), tree reuse + game step + the end / refill of slots (rz_play_apply).  -> the log row."""
        st = self.stream()
        check(self.lib.rz_play_draw(self.handle, st), 'rz_play_draw')
        self.flush_deferred()
        check(self.lib.rz_play_apply(self.handle, st), 'rz_play_apply')
        self.roots_epoch += 1
        row = self.play_steps % self.play_log.shape[0]
        self.play_steps += 1
        return row

    def warm_move_graph(self, evaluator):
        """A hipGraph of one WHOLE move for an evaluator whose search is the one-launch resident kernel (rz_net_search_resident):
        first selection, n_playout simulations, the draw, the policy GEMM + priors of the search, tree reuse + game step + end /
        refill of slots -- replayed once per move (play_move_replay), the log row taken from the device's own step counter.  None
        when the route or the store (fewer than n_playout slots) does not allow it."""
        t = self.torch
        res_ok = getattr(evaluator, 'resident_ok', None)
        if res_ok is None or not res_ok(self) or getattr(self, 'play_log', None) is None:
            return None
        n = self.n_playout
        self.flush_deferred()
        if self._deferred_begin(evaluator, n) < n or self._def_slots < n:   # (reserves the store; a search must fit between two flushes)
            return None
        inner = getattr(evaluator, 'inner', evaluator)
        if getattr(inner, 'resident_delta_ok', None) is not None and inner.resident_delta_ok(self) and getattr(inner.hip, '_delta_games', 0) < self.n_games:
            t.cuda.synchronize(self.device)
            inner.hip.delta_reserve(self.n_games)   # (before the capture: the base cache of the receptive-field trunk)
        hip, lib, h = evaluator.hip, self.lib, self.handle
        t.cuda.synchronize(self.device)
        graph = t.cuda.CUDAGraph()
        self._capturing = True
        try:
            with t.cuda.graph(graph):
                st = self.stream()
                evaluator.search_resident(self, n, True)
                check(lib.rz_play_draw(h, st), 'rz_play_draw')
                logits = hip.deferred_gemm(self.n_leaves, n)
                check(lib.rz_deferred_flush(h, ctypes.byref(logits), n, st), 'rz_deferred_flush')
                check(lib.rz_play_apply(h, st), 'rz_play_apply')
        finally:
            self._capturing = False
        self._move_graph_ev = evaluator
        self._move_graph = graph
        return graph

    def play_move_replay(self, graph):
        """One whole move from the graph of warm_move_graph -> the log row it writes."""
        if graph is not getattr(self, '_move_graph', None):
            raise HipError('this whole-move graph was dropped (%s): call warm_move_graph again' % getattr(self, '_graphs_dropped', 'a buffer it addresses moved'))
        if self._def_pending > 0:
            self.flush_deferred()   # (leaves of eager steps before this move: the graph's own flush covers its n_playout slots from 0)
        graph.replay()
        self.roots_epoch += 1
        row = self.play_steps % self.play_log.shape[0]
        self.play_steps += 1
        return row

    def play_refill(self):
        """rz_play_apply alone: idle slots take games from the queue (the start of a run).  Counts as a move step whose log row
        holds nothing to read (no draw has written it)."""
        self.flush_deferred()
        check(self.lib.rz_play_apply(self.handle, self.stream()), 'rz_play_apply')
        self.roots_epoch += 1
        self.play_steps += 1

    def play_resolve(self, slot, move):
        check(self.lib.rz_play_resolve(self.handle, int(slot), int(move), self.stream()), 'rz_play_resolve')

    def play_stop(self):
        self.flush_deferred()
        check(self.lib.rz_play_stop(self.handle, self.stream()), 'rz_play_stop')
        self._play_active = None   # (the host-driven loop sets active flags of its own again: set_active)

    def play_state(self):
        """-> (game ids int64 [G] (-1 idle), plies, states (0 idle / 1 running / 2 stalled), move steps done); synchronises."""
        gid = np.zeros(self.n_games, np.int64)
        ply = np.zeros(self.n_games, np.int32)
        state = np.zeros(self.n_games, np.int32)
        steps = ctypes.c_int64(0)
        check(self.lib.rz_play_state(self.handle, ctypes.c_void_p(gid.ctypes.data), ctypes.c_void_p(ply.ctypes.data),
                                     ctypes.c_void_p(state.ctypes.data), ctypes.byref(steps)), 'rz_play_state')
        return gid, ply, state, steps.value

    # ------------------------------------------------------------------ inspection
    def arena(self, game=0):
        """Host snapshot of one game's tree.  Per node record (slot): N, W, FC (slot of the first child
        record, -1 = none yet), NV (visited children = child records in use), K (children, 0 = not
        expanded), PB (offset into PRI of the node's K child priors); the visited child r < NV of a node
        is slot FC + r and the prior of ANY child r < K is PRI[PB + r]."""
        self.flush_deferred()
        st = self.stats()
        cap = int(st.arena_slots)
        n = np.zeros(cap, np.int32)
        w = np.zeros(cap, np.float64)
        fc = np.zeros(cap, np.int32)
        nv = np.zeros(cap, np.int32)
        kk = np.zeros(cap, np.int32)
        pb = np.zeros(cap, np.int32)
        pri = np.zeros(int(st.prior_floats), np.float32)
        root_prior = ctypes.c_float(1.0)
        top, ptop = ctypes.c_int32(0), ctypes.c_int32(0)

        def hp(a):
            return ctypes.c_void_p(a.ctypes.data)

        def ref(x):
            return ctypes.cast(ctypes.byref(x), ctypes.c_void_p)

        check(self.lib.rz_copy_arena(self.handle, int(game), cap, hp(n), hp(w), hp(fc), hp(nv), hp(kk), hp(pb),
                                     ref(root_prior), ref(top)), 'rz_copy_arena')
        check(self.lib.rz_copy_priors(self.handle, int(game), pri.size, hp(pri), ref(ptop)), 'rz_copy_priors')
        k = top.value
        return {'N': n[:k], 'W': w[:k], 'FC': fc[:k], 'NV': nv[:k], 'K': kk[:k], 'PB': pb[:k],
                'PRI': pri[:ptop.value], 'root_prior': float(root_prior.value), 'top': k}

    def tree_dump(self, game=0):
        """{path of actions: (N, W)} over visited nodes (+ the root), like the oracle's."""
        ar = self.arena(game)
        stones, _, _ = self.get_roots()
        occ0 = bits_to_int(stones[game, 0]) | bits_to_int(stones[game, 1])
        out = {}
        stack = [((), 0, occ0)]
        while stack:
            path, slot, occ = stack.pop()
            out[path] = (int(ar['N'][slot]), float(ar['W'][slot]))
            nv = int(ar['NV'][slot])
            if int(ar['K'][slot]) == 0 or nv == 0:
                continue
            fc = int(ar['FC'][slot])
            legal = self.legal_actions(occ)
            for r in range(nv):
                if int(ar['N'][fc + r]) > 0:  # PUCT initialises every child; only visited ones count
                    a = legal[r]
                    stack.append((path + (a, ), fc + r, occ | (1 << self.cell_of_action(occ, a))))
        return out

    # ------------------------------------------------------------------ rules helpers (host)
    def legal_actions(self, occ):
        """Ascending legal actions of a position given its occupancy bitboard (Python int)."""
        if self.game == 'connect4':
            top = (self.rows - 1) * self.cols
            return [c for c in range(self.cols) if not (occ >> (top + c)) & 1]
        return [c for c in range(self.n_cells) if not (occ >> c) & 1]

    def cell_of_action(self, occ, action):
        """Cell a move occupies: the cell itself (Gomoku) or the lowest empty cell of the column."""
        if self.game != 'connect4':
            return int(action)
        for row in range(self.rows):
            cell = row * self.cols + int(action)
            if not (occ >> cell) & 1:
                return cell
        raise ValueError('column %d is full' % action)

    def uct_scores(self, w, n, n_parent, c_puct):
        """The select arithmetic alone (for bit-exactness tests)."""
        t = self.torch
        dw = t.from_numpy(np.ascontiguousarray(w, np.float64)).to(self.device)
        dn = t.from_numpy(np.ascontiguousarray(n, np.int32)).to(self.device)
        dp = t.from_numpy(np.ascontiguousarray(n_parent, np.int32)).to(self.device)
        out = t.empty_like(dw)
        check(self.lib.rz_uct_scores(self.handle, _ptr(dw), _ptr(dn), _ptr(dp), float(c_puct),
                                     _ptr(out), dw.numel(), self.stream()), 'rz_uct_scores')
        return out.cpu().numpy()

    def log_table_size(self):
        n = ctypes.c_int64(0)
        check(self.lib.rz_log_table_size(self.handle, ctypes.byref(n)), 'rz_log_table_size')
        return n.value

    def upload_log_table(self, table):
        table = np.ascontiguousarray(table, dtype=np.float64)
        check(self.lib.rz_upload_log_table(self.handle, ctypes.c_void_p(table.ctypes.data), table.size),
              'rz_upload_log_table')
