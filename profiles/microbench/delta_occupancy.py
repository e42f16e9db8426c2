"""Where the workgroups of ONE k_trunk_delta launch ran and when (the launch trace of rz_trace.h): workgroups per CU at a time, the
spread of their lifetimes.   python profiles/microbench/delta_occupancy.py [n_boards ...]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from rlzero_amd import _hip
from rlzero_amd.engine import HipNet, _ptr
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
sys.path.insert(0, 'profiles/microbench')
from delta_trunk_bench import positions

sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512]
torch.manual_seed(0)
hip = HipNet(15, 'cuda:0', max_boards=max(sizes)).load_state_dict(PolicyValueNet(15).state_dict())
rng = np.random.default_rng(1)
for n in sizes:
    hip.deferred_reserve(n, 2)
    hip.delta_reserve(n)
    slot = torch.zeros(n, dtype=torch.int32, device='cuda:0')
    root, rtm, leaf, tm, last = positions(rng, n, 40, 'mix')
    d = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to('cuda:0') for a in (root, rtm, leaf, tm, last)]
    hip.delta_bases(_ptr(d[0]), _ptr(d[1]), n)
    trace = torch.zeros(2 + 4 * 2 * 1 * n, dtype=torch.int64, device='cuda:0')
    trace[0], trace[1] = 1, n
    _hip.check(hip.lib.rz_net_trace_attach(hip.handle, _ptr(trace)), 'attach')
    for _ in range(3):
        hip.delta_leaves(_ptr(d[2]), _ptr(d[3]), _ptr(d[4]), n, slot_of=_ptr(slot))
    torch.cuda.synchronize()
    rec = trace[2:2 + 4 * n].cpu().numpy().reshape(n, 4)
    _hip.check(hip.lib.rz_net_trace_attach(hip.handle, None), 'detach')
    t0, t1, hw = rec[:, 0], rec[:, 1], rec[:, 3]
    xcc, hwid = hw >> 32, hw & 0xffffffff
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    place = xcc * 1000 + se * 100 + sh * 16 + cu
    life = (t1 - t0) / 100.0   # us (100 MHz)
    span = (t1.max() - t0.min()) / 100.0
    places = len(set(place.tolist()))
    # the largest number of workgroups alive together on one place
    worst = 0
    for p in set(place.tolist()):
        idx = np.nonzero(place == p)[0]
        ev = sorted([(t0[i], 1) for i in idx] + [(t1[i], -1) for i in idx])
        c = 0
        for _, s_ in ev:
            c += s_
            worst = max(worst, c)
    print('%4d boards: launch span %.1f us, workgroup lifetime %.1f / %.1f / %.1f us (min / median / max), %d distinct CUs, at most %d workgroups alive on one CU, '
          'starts within %.1f us' % (n, span, life.min(), np.median(life), life.max(), places, worst, (t0.max() - t0.min()) / 100.0), flush=True)
