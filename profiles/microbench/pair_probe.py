import sys, os
sys.path.insert(0, '/root/repo')
import torch
from rlzero_amd.engine import HipNet
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
torch.manual_seed(0)
def t(shape, n, reps=200):
    rows, cols = shape
    net = PolicyValueNet(rows, cols, rows * cols)
    hip = HipNet((rows, cols, rows * cols), 'cuda:0', max_boards=n).load_state_dict(net.state_dict())
    obs = (torch.rand(n, 4, rows, cols, device='cuda:0') > 0.5).float()
    for _ in range(20): hip.trunk_internal(obs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): hip.trunk_internal(obs)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    hip.close()
    return us
for shape, n in (((6, 7), 256), ((6, 7), 512), ((13, 7), 256), ((6, 15), 256), ((9, 9), 256), ((12, 7), 256), ((13,7),128), ((6,7),128)):
    print(shape, n, 'boards: %.2f us per launch' % t(shape, n))
