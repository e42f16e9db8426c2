"""Per-phase shader-clock cycles of k_trunk_split per board (development aid): needs a -DRZ_NET_PROFILE build next to this file."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import rlzero_amd._hip as H
H.library_path = lambda: os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librlzero_netprof.so')
import numpy as np, torch
from rlzero_amd.engine import HipNet
from oracle.evaluators import numpy_weights
lib = H.load()
names = ['conv1', 'bar', 'conv2 loop', 'conv2 epi', 'bar', 'conv3 loop', 'heads epi', 'stores', '-', 'prologue', 'kernel']
for B, boards in ((15, 768), (15, 256), (9, 64), (9, 1024), (6, 256), (3, 1)):
    bs = B if isinstance(B, tuple) else (B, B)
    w = numpy_weights(B, 1)
    net = HipNet(B, 'cuda:0', boards).load_state_dict(w)
    x = (torch.rand(boards, 4, bs[0], bs[1], device='cuda:0') < 0.3).float()
    for _ in range(3):
        net.trunk_internal(x)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.rz_net_debug_profile(out)
    v = list(out)
    per = max(1, (boards + 255) // 256)
    print('board %s, %d boards (%d per workgroup): kernel %d cycles; per board: ' % (str(B), boards, per, v[10]) +
          '  '.join('%s=%d' % (n, x / per) for n, x in zip(names[:8], v[:8])) + '  | prologue=%d (issue loads %d, zero %d, barrier %d, stores %d, barrier %d)' % (v[9], v[11], v[12], v[13], v[14], v[15]))
