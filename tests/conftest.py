import gzip
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(HERE, 'golden')

# rlzero_amd claims 8 hardware queues on import (four lanes of games: rlzero_amd/__init__.py); the collection hook below asks
# torch for the GPU -- which starts the runtime -- before any test imports the package, so the variable is set here
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need an MI355X and the built extension: skipped elsewhere, so a plain `pytest` is green on a
    CPU host (the driver's GPU tier selects them with -m gpu on a GPU box, where they run)."""
    reason = None
    try:
        import torch
        if not torch.cuda.is_available():
            reason = 'no GPU in this container (run on the GPU box: pytest -m gpu)'
    except Exception as exc:  # noqa: BLE001
        reason = 'torch unavailable: %r' % (exc, )
    if reason is None and not os.path.exists(os.path.join(REPO, 'rlzero_amd', 'librlzero_hip.so')):
        reason = 'rlzero_amd/librlzero_hip.so is not built (python -m rlzero_amd._build)'
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for item in items:
            if 'gpu' in item.keywords:
                item.add_marker(skip)


def load_golden_json(name):
    with gzip.open(os.path.join(GOLDEN, name + '.gz'), 'rb') as f:
        return json.loads(f.read().decode())


def load_golden_npz(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope='session')
def g1():
    return load_golden_json('g1_rules.json')


@pytest.fixture(scope='session')
def g2():
    return load_golden_json('g2_search.json')


@pytest.fixture(scope='session')
def g3():
    return load_golden_json('g3_games.json')


@pytest.fixture(scope='session')
def g2net():
    return load_golden_json('g2_netleaf.json')


@pytest.fixture(scope='session')
def g4():
    return load_golden_npz('g4_net.npz')


@pytest.fixture(scope='session')
def g5():
    return load_golden_npz('g5_equi.npz')


def unhex(x):
    return float.fromhex(x)


def bits_of_planes(obs):
    """[4,B,B] array of 0/1 -> list of 4 hex strings (bit h*B+w), as in the fixtures."""
    out = []
    for p in range(4):
        flat = np.asarray(obs[p]).reshape(-1)
        assert set(np.unique(flat)) <= {0.0, 1.0}
        out.append(hex(sum(1 << i for i, v in enumerate(flat) if v == 1.0)))
    return out


@pytest.fixture(scope='session')
def g6():
    return load_golden_json('g6_rollout.json')


@pytest.fixture(scope='session')
def g7():
    return load_golden_json('g7_train.json')
