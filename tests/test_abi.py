"""The C-ABI library builds, loads on a GPU-less host and exports every symbol that
include/rlzero_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest
from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, 'include', 'rlzero_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rz_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib_path():
    from rlzero_amd import _build
    return _build.build(verbose=False)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 36
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_the_header(lib_path):
    from rlzero_amd import _hip
    assert _hip.exported_symbols() == _declared()
    lib = _hip.load()
    assert lib.rz_abi_version() == _hip.ABI_VERSION
    header = open(os.path.join(REPO, 'include', 'rlzero_hip.h')).read()
    assert '#define RZ_ABI_VERSION %d' % _hip.ABI_VERSION in header


def test_config_struct_layout_matches_header():
    from rlzero_amd import _hip
    # 8 int32, 2 doubles, 6 int32 -> 72 bytes, doubles 8-aligned at offset 32
    assert ctypes.sizeof(_hip.RzConfig) == 72
    assert _hip.RzConfig.c_puct.offset == 32 and _hip.RzConfig.device.offset == 48
    assert ctypes.sizeof(_hip.RzStats) == 64
    # rz_mz_cartpole_play: 4 pointers, 2 uint64, 3 doubles, pointer, 2 int32, int64, pointer, int64, 2 pointers, int64
    assert ctypes.sizeof(_hip.RzMzCartPolePlay) == 136 and _hip.RzMzCartPolePlay.first_step.offset == 88
    # rz_value_head: five pointers + two int32; rz_deferred_logits: a pointer + two int32 (the deferred-priors route, ABI 22)
    assert ctypes.sizeof(_hip.RzValueHead) == 48 and _hip.RzValueHead.ld.offset == 40 and _hip.RzValueHead.groups.offset == 44
    assert ctypes.sizeof(_hip.RzDeferredLogits) == 16 and _hip.RzDeferredLogits.rows_per_slot.offset == 12
    assert _hip.RzMzCartPolePlay.max_entries.offset == 128
    # rz_play_config: uint64, two doubles, three pointers, two int32 (the move step on the device, ABI 24)
    assert ctypes.sizeof(_hip.RzPlayConfig) == 56 and _hip.RzPlayConfig.d_queue_ids.offset == 24 and _hip.RzPlayConfig.ring_steps.offset == 48


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: constructing an engine on a host without a GPU must raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from rlzero_amd.engine import MCTSEngine
    from rlzero_amd._hip import HipError
    with pytest.raises(HipError):
        MCTSEngine(3, 3, device='cuda:0')
    with pytest.raises(HipError):
        MCTSEngine(3, 3, device='cpu')


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import rlzero_amd, rlzero_amd.engine, rlzero_amd.mcts, "
            "rlzero_amd.games, rlzero_amd.selfplay, rlzero; "
            "bad=[m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; "
            "assert not bad, bad" % REPO)
    subprocess.run([sys.executable, '-c', code], check=True)
    for root, _, files in os.walk(os.path.join(REPO, 'rlzero_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(root, f)).read()
                # mentions in comments are fine; importing / including / loading is not
                assert not re.search(r'^\s*(import|from)\s+oracle\b|#include\s*[<"].*oracle|oracle/|oracle\.', text,
                                     flags=re.M), f
