import ctypes, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rlzero_amd import _hip
from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
lib = _hip.load()
T = {0: 'sel: first loads', 1: 'sel: level, fresh child', 2: 'sel: level, scan', 6: 'sel: legal_of', 3: 'sel: nth + set', 7: 'sel: after loop', 4: 'sel: terminal rule', 5: 'sel: stores',
     10: 'eb: loads + value', 11: 'eb: expand', 12: 'eb: backup'}
for B, n_row, games, sims in ((3, 3, 1, 25), (9, 5, 64, 200), (15, 5, 128, 800)):
    torch.manual_seed(0)
    net = PolicyValueNet(B).to('cuda:0')
    ev = HipNetEvaluator(net, B, 'cuda:0', max_boards=games)
    eng = MCTSEngine(B, n_row, n_games=games, n_playout=sims, device='cuda:0', add_noise=True)
    eng.reset_games()
    eng.sim_chunk(ev, sims)
    out0 = (ctypes.c_longlong * 40)(); lib.rz_net_debug_profile(out0)
    # a second search on the grown tree (what a move with reuse sees is in between)
    eng.sim_chunk(ev, sims)
    out = (ctypes.c_longlong * 40)(); lib.rz_net_debug_profile(out)
    d = [out[24 + i] - out0[24 + i] for i in range(16)]
    nsel, nlev = d[9], d[8]
    print('%dx%d: selections %d, levels %d (%.2f per selection)' % (B, B, nsel, nlev, nlev / max(nsel, 1)))
    print('   ' + '  '.join('%s=%d' % (T[k], d[k] / max(nsel, 1)) for k in sorted(T)), ' (cycles per simulation)')
    eng.close(); ev.hip.close()
