"""Go / no-go for receptive-field leaf evaluation (task 1): k_trunk_delta against k_trunk_rows on the leaves a 15 x 15 /
800 search meets -- the root plus one stone anywhere (28 % of a move's leaves), the root plus that stone plus one of the FIRST legal
cells (72 %: unvisited children are taken in action order, node.py:41-42, so a root child's first children are cells 0, 1, 2 ..).
Launch durations from HIP events on one stream, the deferred route's outputs (store slot 0 + value rows) in both cases.
    python profiles/microbench/delta_trunk_bench.py [n_boards ...]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from rlzero_amd.engine import HipNet, _ptr   # noqa: E402
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet   # noqa: E402


def positions(rng, n, stones, kind):
    S = 225
    root = np.zeros((n, 2, 4), np.uint64)
    leaf = np.zeros((n, 2, 4), np.uint64)
    tm = np.zeros(n, np.int32)
    last = np.zeros(n, np.int32)
    for i in range(n):
        seq = list(rng.permutation(S)[:stones])
        free = [c for c in range(S) if c not in set(seq)]
        d = 1 if kind == 'd1' else 2 if kind == 'd2' else (1 if rng.random() < 0.28 else 2)
        a = int(rng.choice(free))
        add = [a]
        if d == 2:
            rest = [c for c in free if c != a]
            add.append(rest[int(rng.integers(0, 4))])   # one of the first legal cells
        for j, c in enumerate(seq):
            root[i, j % 2, c >> 6] |= np.uint64(1) << np.uint64(c & 63)
        leaf[i] = root[i]
        for j, c in enumerate(add):
            leaf[i, (stones + j) % 2, c >> 6] |= np.uint64(1) << np.uint64(c & 63)
        tm[i] = (stones + d) % 2
        last[i] = add[-1]
    return root, np.full(n, stones % 2, np.int32), leaf, tm, last


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512]
    torch.manual_seed(0)
    hip = HipNet(15, 'cuda:0', max_boards=max(sizes)).load_state_dict(PolicyValueNet(15).state_dict())
    rng = np.random.default_rng(1)
    for n in sizes:
        hip.deferred_reserve(n, 2)
        hip.delta_reserve(n)
        slot = torch.zeros(n, dtype=torch.int32, device='cuda:0')
        for kind in ('d1', 'd2', 'mix'):
            root, rtm, leaf, tm, last = positions(rng, n, 40, kind)
            d = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to('cuda:0') for a in (root, rtm, leaf, tm, last)]

            class Eng(object):
                n_leaves = n
                def leaf_buffers(self):
                    return _ptr(d[2]), _ptr(d[3]), _ptr(d[4])
                def deferred_slot_ptr(self):
                    return _ptr(slot)
            eng = Eng()
            hip.delta_bases(_ptr(d[0]), _ptr(d[1]), n)

            def timed(fn, reps=40):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) * 1e3 / reps
            full = timed(lambda: hip.trunk_leaves_deferred(eng))
            hip.delta_stats(reset=True)
            delta = timed(lambda: hip.delta_leaves(_ptr(d[2]), _ptr(d[3]), _ptr(d[4]), n, slot_of=_ptr(slot)))
            st = hip.delta_stats()
            nobase = timed(lambda: hip.delta_leaves(_ptr(d[2]), _ptr(d[3]), _ptr(d[4]), n, slot_of=_ptr(slot), without_base=True), reps=10)
            bases = timed(lambda: hip.delta_bases(_ptr(d[0]), _ptr(d[1]), n), reps=10)
            leaves = max(1, st['delta'] + st['no_base'])
            print('%4d boards %-3s  full %7.2f us  delta %7.2f us (%.2fx; %.2f conv3 tiles, %.2f changed cells per leaf, %d without a base)  '
                  'four passes %7.2f us  bases (2 per game) %7.2f us' % (n, kind, full, delta, full / delta, st['tiles3'] / leaves, st['cells'] / leaves,
                                                                       st['no_base'], nobase, bases), flush=True)


if __name__ == '__main__':
    main()
