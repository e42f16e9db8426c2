"""Per-phase shader-clock cycles of k_trunk_split / k_trunk_rows per board (development aid): needs a -DRZ_NET_PROFILE build next to
this file.  k_trunk_rows: 'stores' = barrier + the four waves' shares summed + feature stores, 'end bar' = the barrier behind them."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import rlzero_amd._hip as H
H.library_path = lambda: os.environ.get('RZ_NETPROF_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librlzero_netprof.so')
import numpy as np, torch
from rlzero_amd.engine import HipNet
from oracle.evaluators import numpy_weights
lib = H.load()
planes_route = 'planes' in sys.argv[1:]
names = ['conv1', 'bar', 'conv2 loop', 'conv2 epi', 'bar', 'conv3 loop', 'heads epi', 'stores', 'end bar', 'prologue', 'kernel']
CASES = [(15, 768, 'split_f16'), (15, 768, 'split_f16_tiles'), (15, 256, 'split_f16'), (15, 256, 'split_f16_tiles')]
if 'all' in sys.argv[1:]:
    CASES += [(9, 64, 'split_f16'), (9, 1024, 'split_f16'), (6, 256, 'split_f16'), (3, 1, 'split_f16')]
for B, boards, algo in CASES:
    bs = B if isinstance(B, tuple) else (B, B)
    w = numpy_weights(B, 1)
    net = HipNet(B, 'cuda:0', boards).load_state_dict(w).set_algo(algo)
    x = (torch.rand(boards, 4, bs[0], bs[1], device='cuda:0') < 0.3).float()
    # the production route: leaf bitboards (random disjoint stones of two colours), side to move, last cell
    rs = np.random.RandomState(0)
    cells = bs[0] * bs[1]
    occ = rs.rand(boards, cells)
    stones = np.zeros((boards, 2, 4), dtype=np.uint64)
    for c, (lo, hi) in enumerate(((0.0, 0.2), (0.2, 0.4))):
        for cell in range(cells):
            m = (occ[:, cell] >= lo) & (occ[:, cell] < hi)
            stones[m, c, cell >> 6] |= np.uint64(1) << np.uint64(cell & 63)
    d_stones = torch.from_numpy(stones.view(np.int64)).to('cuda:0')
    d_tm = torch.from_numpy(rs.randint(0, 2, boards).astype(np.int32)).to('cuda:0')
    d_last = torch.from_numpy(rs.randint(0, cells, boards).astype(np.int32)).to('cuda:0')
    net.reserve(boards)
    for _ in range(3):
        if planes_route:
            net.trunk_internal(x)
        else:
            H.check(lib.rz_net_trunk_leaves(net.handle, d_stones.data_ptr(), d_tm.data_ptr(), d_last.data_ptr(), boards, None), 'trunk_leaves')
    torch.cuda.synchronize()
    # wall time per launch of the same (instrumented) kernel -> the shader clock it ran at = cycles / time
    reps = 200
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        if planes_route:
            net.trunk_internal(x)
        else:
            H.check(lib.rz_net_trunk_leaves(net.handle, d_stones.data_ptr(), d_tm.data_ptr(), d_last.data_ptr(), boards, None), 'trunk_leaves')
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / reps
    out = (ctypes.c_longlong * 16)()
    lib.rz_net_debug_profile(out)
    v = list(out)
    per = max(1, (boards + 255) // 256)
    print('board %s %s, %d boards (%d per workgroup): kernel %d cycles in %.1f us back to back = %.2f GHz; per board: ' % (str(B), algo, boards, per, v[10], us, v[10] / us / 1e3) +
          '  '.join('%s=%d' % (n, x / per) for n, x in zip(names[:9], v[:9])) + '  | prologue=%d (issue loads %d, zero %d, barrier %d, stores %d, barrier %d)' % (v[9], v[11], v[12], v[13], v[14], v[15]))
