"""ctypes binding of include/rlzero_hip.h (the C ABI of the HIP engine).

There is deliberately NO fallback: if ``librlzero_hip.so`` is missing or a call fails the
product raises.  Build it with ``python -m rlzero_amd._build``.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_uint64,
                    c_void_p)

ABI_VERSION = 26
BOARD_WORDS = 4
MAX_BOARD_SIZE = 16
MAX_IN_FLIGHT = 16

OK = 0
SCORE_UCT_REF, SCORE_PUCT = 0, 1
GAME_GOMOKU, GAME_CONNECT4 = 0, 1
NET_DIRECT, NET_WINOGRAD_F4, NET_SPLIT_F16, NET_SPLIT_F16_TILES, NET_SPLIT_F16_FP8 = 0, 1, 2, 3, 4
NET_FLAG_F16_RANGE = 1
NET_HEADS_AUTO, NET_HEADS_F32, NET_HEADS_SPLIT_32, NET_HEADS_SPLIT_64, NET_HEADS_SPLIT_PARTS, NET_HEADS_IN_TRUNK = 0, 1, 2, 3, 4, 5
EVAL_V0, EVAL_VLIN = 0, 1
FLAG_NAMES = {1: 'arena full', 2: 'block queue full', 4: 'illegal move', 8: 'ln table too short',
              16: 'internal'}
FLAG_REUSE_DROPPED = 32  # not an error: a kept subtree exceeded the arena's carry limit and was dropped (counted)


class RzConfig(Structure):
    _fields_ = [('abi_version', c_int32), ('game_kind', c_int32), ('board_size', c_int32),
                ('n_in_row', c_int32), ('n_games', c_int32), ('n_playout', c_int32),
                ('score_mode', c_int32), ('add_noise', c_int32), ('c_puct', c_double),
                ('pool_factor', c_double), ('device', c_int32), ('noise_seed', c_int32),
                ('board_height', c_int32), ('board_width', c_int32),
                ('sims_in_flight', c_int32), ('in_flight_impl', c_int32)]


class RzStats(Structure):
    _fields_ = [('error_flags', c_int32), ('first_bad_game', c_int32), ('arena_slots', c_int64),
                ('prior_floats', c_int64), ('max_slots_used', c_int64), ('max_blocks_used', c_int64),
                ('device_bytes', c_int64), ('n_select_calls', c_int64), ('reuse_dropped', c_int64)]


class RzMzConfig(Structure):
    _fields_ = [('abi_version', c_int32), ('n_games', c_int32), ('n_actions', c_int32), ('n_sims', c_int32),
                ('discount', c_double), ('pb_c_base', c_double), ('pb_c_init', c_double),
                ('device', c_int32), ('reserved', c_int32)]


class RzMzCartPolePlay(Structure):
    """rz_mz_cartpole_play: environments, random streams and the device-side history of rz_mz_play_cartpole."""
    _fields_ = [('d_state', c_void_p), ('d_steps', c_void_p), ('d_episode', c_void_p), ('d_episode_start', c_void_p),
                ('env_seed', c_uint64), ('noise_seed', c_uint64), ('noise_frac', c_double), ('dirichlet_alpha', c_double),
                ('temperature', c_double), ('d_ring', c_void_p), ('ring_steps', c_int32), ('reserved', c_int32),
                ('first_step', c_int64), ('d_arena', c_void_p), ('arena_rows', c_int64), ('d_counters', c_void_p),
                ('d_entries', c_void_p), ('max_entries', c_int64)]


class RzRawHeads(Structure):
    """rz_raw_heads: the FC GEMM's outputs as the tree kernels consume them (device pointers)."""
    _fields_ = [('raw', c_void_p), ('hid', c_void_p), ('w2', c_void_p), ('b2', c_void_p), ('act_scale', c_void_p),
                ('act_bias', c_void_p), ('val_scale', c_void_p), ('val_bias', c_void_p), ('raw_part_stride', c_int64),
                ('hid_part_stride', c_int64), ('ld', c_int32), ('n_parts', c_int32)]


class RzValueHead(Structure):
    """rz_value_head: what the trunk of the deferred-priors route leaves for the tree step (device pointers)."""
    _fields_ = [('valfeat', c_void_p), ('w1t', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p),
                ('ld', c_int32), ('groups', c_int32)]


class RzDeferredLogits(Structure):
    """rz_deferred_logits: the policy logits of the stored leaves after rz_net_deferred_gemm."""
    _fields_ = [('raw', c_void_p), ('ld', c_int32), ('rows_per_slot', c_int32)]


class RzPlayConfig(Structure):
    """rz_play_config: the move step on the device (uniform / noise seed, temperature, the shared queue of game ids, the log)."""
    _fields_ = [('seed', c_uint64), ('temperature', c_double), ('stall_margin', c_double), ('d_queue_ids', c_void_p),
                ('d_queue_ctl', c_void_p), ('d_log', c_void_p), ('ring_steps', c_int32), ('reserved', c_int32)]


PLAY_RECORD_WORDS = 8
PLAY_RUNNING, PLAY_STALLED, PLAY_RESOLVED, PLAY_ENDED, PLAY_SEARCHED = 1, 2, 4, 8, 16


class HipError(RuntimeError):
    pass


P = c_void_p  # device / host pointers and streams travel as integers

_SIGNATURES = {
    'rz_abi_version': (c_int, []),
    'rz_last_error': (c_char_p, []),
    'rz_source_hash': (c_char_p, []),
    'rz_create': (c_int, [POINTER(RzConfig), POINTER(c_void_p)]),
    'rz_destroy': (c_int, [P]),
    'rz_geometry': (c_int, [P, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    'rz_upload_log_table': (c_int, [P, P, c_int64]),
    'rz_log_table_size': (c_int, [P, POINTER(c_int64)]),
    'rz_set_roots': (c_int, [P, P, P, P, P, c_int, P]),
    'rz_get_roots': (c_int, [P, P, P, P, P]),
    'rz_set_active': (c_int, [P, P, P]),
    'rz_set_noise_keys': (c_int, [P, P, P, P]),
    'rz_select_step': (c_int, [P, P, P]),
    'rz_set_in_flight': (c_int, [P, c_int32, c_int32]),
    'rz_leaf_buffers': (c_int, [P, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    'rz_encode_leaf_obs': (c_int, [P, P, P]),
    'rz_encode_root_obs': (c_int, [P, P, P]),
    'rz_get_leaves': (c_int, [P, P, P, P, P, P]),
    'rz_eval_synthetic': (c_int, [P, c_int, P, P, P]),
    'rz_eval_rollout': (c_int, [P, c_uint64, c_uint32, c_int32, P, P]),
    'rz_expand_backup': (c_int, [P, P, P, P]),
    'rz_expand_backup_f64': (c_int, [P, P, P, P]),
    'rz_expand_backup_probs': (c_int, [P, P, P, P]),
    'rz_tree_step': (c_int, [P, P, P, P, P]),
    'rz_expand_backup_raw': (c_int, [P, POINTER(RzRawHeads), P]),
    'rz_tree_step_raw': (c_int, [P, POINTER(RzRawHeads), P, P]),
    'rz_trace_attach': (c_int, [P, P]),
    'rz_net_trace_attach': (c_int, [P, P]),
    'rz_device_view': (c_int, [P, P, c_int64]),
    'rz_deferred_reserve': (c_int, [P, c_int32]),
    'rz_deferred_slots': (c_int, [P, POINTER(c_void_p)]),
    'rz_expand_backup_deferred': (c_int, [P, POINTER(RzValueHead), P]),
    'rz_tree_step_deferred': (c_int, [P, POINTER(RzValueHead), P]),
    'rz_deferred_flush': (c_int, [P, POINTER(RzDeferredLogits), c_int32, P]),
    'rz_root_visits': (c_int, [P, P, P]),
    'rz_root_wsum': (c_int, [P, P, P]),
    'rz_root_priors': (c_int, [P, P, P]),
    'rz_root_stats': (c_int, [P, P, P, P]),
    'rz_advance_roots': (c_int, [P, P, P]),
    'rz_step_games': (c_int, [P, P, P, P, P]),
    'rz_play_attach': (c_int, [P, POINTER(RzPlayConfig)]),
    'rz_play_draw': (c_int, [P, P]),
    'rz_play_apply': (c_int, [P, P]),
    'rz_play_resolve': (c_int, [P, c_int32, c_int32, P]),
    'rz_play_stop': (c_int, [P, P]),
    'rz_play_state': (c_int, [P, P, P, P, POINTER(c_int64)]),
    'rz_get_stats': (c_int, [P, POINTER(RzStats)]),
    'rz_clear_errors': (c_int, [P]),
    'rz_poll_errors': (c_int, [P, POINTER(c_int32), POINTER(c_int32), P]),
    'rz_copy_arena': (c_int, [P, c_int32, c_int64, P, P, P, P, P, P, P, P]),
    'rz_copy_priors': (c_int, [P, c_int32, c_int64, P, P]),
    'rz_uct_scores': (c_int, [P, P, P, P, c_double, P, c_int64, P]),
    'rz_net_create': (c_int, [c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    'rz_net_destroy': (c_int, [P]),
    'rz_net_set_algo': (c_int, [P, c_int32]),
    'rz_net_set_max_workgroups': (c_int, [P, c_int32]),
    'rz_net_set_heads_algo': (c_int, [P, c_int32]),
    'rz_net_error_flags': (c_int, [P, POINTER(ctypes.c_uint32)]),
    'rz_net_range_info': (c_int, [P, POINTER(ctypes.c_float)]),
    'rz_net_load': (c_int, [P, POINTER(c_void_p), c_int32]),
    'rz_net_reserve': (c_int, [P, c_int32]),
    'rz_net_trunk': (c_int, [P, P, c_int32, P, P]),
    'rz_net_trunk_leaves': (c_int, [P, P, P, P, c_int32, P]),
    'rz_net_deferred_reserve': (c_int, [P, c_int32, c_int32]),
    'rz_net_trunk_leaves_deferred': (c_int, [P, P, P, P, c_int32, P, POINTER(RzValueHead), P]),
    'rz_net_deferred_gemm': (c_int, [P, c_int32, c_int32, POINTER(RzDeferredLogits), P]),
    'rz_net_search_resident': (c_int, [P, P, c_int32, c_int32, P]),
    'rz_net_delta_reserve': (c_int, [P, c_int32]),
    'rz_net_delta_invalidate': (c_int, [P, P]),
    'rz_net_delta_resident': (c_int, [P, c_int32]),
    'rz_net_delta_bases': (c_int, [P, P, P, c_int32, P]),
    'rz_net_delta_leaves': (c_int, [P, P, P, P, c_int32, P, P, P, c_int32, POINTER(RzValueHead), P]),
    'rz_net_delta_stats': (c_int, [P, POINTER(ctypes.c_uint32), c_int32]),
    'rz_net_delta_bases_engine': (c_int, [P, P, P]),
    'rz_net_delta_step': (c_int, [P, P, POINTER(RzValueHead), P]),
    'rz_net_delta_trunk_engine': (c_int, [P, P, P]),
    'rz_net_heads': (c_int, [P, c_int32, P, P, P]),
    'rz_net_heads_gemm': (c_int, [P, c_int32, POINTER(RzRawHeads), P]),
    'rz_net_forward': (c_int, [P, P, c_int32, P, P, P]),
    'rz_mz_create': (c_int, [POINTER(RzMzConfig), POINTER(c_void_p)]),
    'rz_mz_destroy': (c_int, [P]),
    'rz_mz_upload_log_table': (c_int, [P, P, c_int64]),
    'rz_mz_init_roots': (c_int, [P, P, P, c_double, P, P]),
    'rz_mz_select': (c_int, [P, P, P, P, P, P]),
    'rz_mz_expand_backup': (c_int, [P, P, P, P, P, P]),
    'rz_mz_load_model': (c_int, [P, POINTER(c_void_p), c_int32, c_int32]),
    'rz_mz_search': (c_int, [P, P, c_int32, P, P, P, P, P, P, P]),
    'rz_mz_set_search_shape': (c_int, [P, c_int32]),
    'rz_mz_load_representation': (c_int, [P, POINTER(c_void_p), c_int32, c_int32, c_int32]),
    'rz_mz_play_cartpole': (c_int, [P, P, c_int32, c_int32, POINTER(RzMzCartPolePlay), P]),
    'rz_cartpole_step': (c_int, [P, P, P, P, c_int32, c_uint64, P, P, P, P, P]),
    'rz_mz_root_children': (c_int, [P, c_int32, P, P]),
    'rz_mz_root_stats': (c_int, [P, P, P, P, P, P]),
    'rz_mz_geometry': (c_int, [P, POINTER(c_int32), POINTER(c_int64)]),
    'rz_mz_error_flags': (c_int, [P, POINTER(c_int32)]),
}

_lib = None


def library_path():
    """The in-tree build; RZ_HIP_LIBRARY names another build of the same ABI (A / B runs of a kernel change)."""
    return os.environ.get('RZ_HIP_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librlzero_hip.so')


def load():
    """Load the shared library (once) and declare every prototype of the header."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HipError('%s is missing: the HIP extension is required (no CPU fallback). '
                       'Build it with `python -m rlzero_amd._build`.' % path)
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.rz_abi_version()
    if got != ABI_VERSION:
        raise HipError('librlzero_hip.so has ABI %d, binding expects %d: rebuild' % (got, ABI_VERSION))
    if not os.environ.get('RZ_HIP_LIBRARY'):   # the in-tree build must be the build OF this tree (no stale binary, whatever its file time)
        from . import _build
        built, tree = (lib.rz_source_hash() or b'').decode(), _build.source_hash()
        if built != tree:
            raise HipError('librlzero_hip.so was built from other sources (hash %s, this tree %s): python -m rlzero_amd._build' % (built, tree))
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what=''):
    if rc != OK:
        msg = load().rz_last_error()
        raise HipError('%s failed (%d): %s' % (what or 'rz call', rc, (msg or b'').decode()))
