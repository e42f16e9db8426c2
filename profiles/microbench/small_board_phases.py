"""Cycles per phase of ONE small-board trunk workgroup (k_trunk_split, a lane's launch from float planes; wave 0 of workgroup 0), from a
library built with -DRZ_NET_PROFILE (see resident_phases.py for the build line):
    RZ_HIP_LIBRARY=scratch/librz_prof.so python profiles/microbench/small_board_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlzero_amd import _hip
from rlzero_amd.engine import HipNet
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
lib = _hip.load()
NAMES = {11: 'pro:loads issued', 12: 'pro:zeroing', 13: 'pro:bar', 14: 'pro:store obs', 15: 'pro:bar', 0: 'conv1', 1: 'bar', 2: 'conv2 loop', 3: 'conv2 epi', 4: 'bar',
         5: 'conv3 loop', 6: 'heads epi', 7: 'stores', 8: 'end bar'}
torch.manual_seed(0)
for shape, n in (((6, 7), 256), ((6, 7), 128), ((9, 9), 256), ((3, 3), 256)):
    rows, cols = shape
    net = PolicyValueNet(rows, cols, rows * cols)
    hip = HipNet((rows, cols, rows * cols), 'cuda:0', max_boards=n).load_state_dict(net.state_dict())
    obs = (torch.rand(n, 4, rows, cols, device='cuda:0') > 0.5).float()
    for _ in range(20): hip.trunk_internal(obs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    a.record()
    for _ in range(reps): hip.trunk_internal(obs)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    hip.trunk_internal(obs); torch.cuda.synchronize()
    out = (ctypes.c_longlong * 24)()
    assert lib.rz_net_debug_profile(out) == 0
    print(shape, n, 'boards: %.2f us per launch; kernel total %d cycles, prologue %d' % (us, out[10], out[9]))
    print('   ' + '  '.join('%s=%d' % (NAMES[k], out[k]) for k in (11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8)))
    hip.close()
