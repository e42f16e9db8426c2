"""Cycles per phase of ONE simulation of the resident search (wave 0 of workgroup 0), from a library built with -DRZ_NET_PROFILE:

    hipcc <flags of rlzero_amd/_build.py> -DRZ_NET_PROFILE -shared -Iinclude rlzero_amd/csrc/*.hip -o scratch/librz_prof.so
    RZ_HIP_LIBRARY=scratch/librz_prof.so python profiles/microbench/resident_phases.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlzero_amd import _hip
from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
lib = _hip.load()
NAMES = {0: 'conv1', 1: 'bar', 2: 'conv2 loop', 3: 'conv2 epi', 4: 'bar', 5: 'conv3 loop', 6: 'heads epi', 7: 'stores', 8: 'end bar',
         16: 'value layer', 17: 'expand+backup', 18: 'select', 19: 'planes'}
for B, n_row, games, sims in ((3, 3, 1, 25), (9, 5, 64, 200), (15, 5, 128, 800)):
    torch.manual_seed(0)
    net = PolicyValueNet(B).to('cuda:0')
    ev = HipNetEvaluator(net, B, 'cuda:0', max_boards=games)
    eng = MCTSEngine(B, n_row, n_games=games, n_playout=sims, device='cuda:0', add_noise=True)
    eng.reset_games()
    assert ev.resident_ok(eng)
    eng.sim_chunk(ev, sims)   # warm
    eng.reset_games()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.sim_chunk(ev, sims); b.record(); torch.cuda.synchronize()
    out = (ctypes.c_longlong * 24)()
    assert lib.rz_net_debug_profile(out) == 0
    per = {k: out[k] / sims for k in NAMES}
    total = out[10] / sims
    print('%dx%d, %d games, %d sims: %.2f us per simulation (events), kernel %d cycles per simulation (%.2f GHz); prologue %d' % (
        B, B, games, sims, 1e3 * a.elapsed_time(b) / sims, total, total / (1e3 * a.elapsed_time(b) / sims) / 1e3, out[9]))
    print('   ' + '  '.join('%s=%d' % (NAMES[k], per[k]) for k in sorted(NAMES)))
    eng.close(); ev.hip.close()
