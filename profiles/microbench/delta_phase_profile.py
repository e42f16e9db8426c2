"""Per-phase shader-clock cycles of k_trunk_delta for the leaf of workgroup 0 (development aid): needs a -DRZ_NET_PROFILE build of the
library next to this file (librlzero_netprof.so: the hipcc lines of rlzero_amd/_build.py plus the define)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import rlzero_amd._hip as H
H.library_path = lambda: os.environ.get('RZ_NETPROF_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librlzero_netprof.so')
import numpy as np, torch
import rlzero_amd._build as B
B.needs_build = lambda: False
from rlzero_amd.engine import HipNet, _ptr
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from delta_trunk_bench import positions
lib = H.load()
names = ['setup', 'dist+ballots', 'bar', 'maps+gather', 'bar', 'conv1', 'bar', 'conv2', 'bar', 'conv3+heads', 'bar', 'features']
torch.manual_seed(0)
n = 128
hip = HipNet(15, 'cuda:0', max_boards=n).load_state_dict(PolicyValueNet(15).state_dict())
hip.deferred_reserve(n, 2)
hip.delta_reserve(n)
slot = torch.zeros(n, dtype=torch.int32, device='cuda:0')
rng = np.random.default_rng(1)
for kind in ('d1', 'd2'):
    root, rtm, leaf, tm, last = positions(rng, n, 40, kind)
    d = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to('cuda:0') for a in (root, rtm, leaf, tm, last)]
    hip.delta_bases(_ptr(d[0]), _ptr(d[1]), n)
    for boards in (1, n):
        for _ in range(3):
            hip.delta_leaves(_ptr(d[2]), _ptr(d[3]), _ptr(d[4]), boards, slot_of=_ptr(slot))
        torch.cuda.synchronize()
        out = (ctypes.c_longlong * 24)()
        lib.rz_net_debug_profile(out)
        v = list(out)
        print('%s, %3d boards in the launch: leaf 0 has %d conv3 tiles; %d cycles: ' % (kind, boards, v[22], v[23]) + '  '.join('%s=%d' % (nm, x) for nm, x in zip(names, v[:12])), flush=True)
