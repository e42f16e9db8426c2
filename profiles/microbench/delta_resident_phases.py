"""Cycles per phase of ONE simulation of the resident search with the receptive-field trunk (k_delta_res; wave 0 of the workgroup of
game 0), from a library built with -DRZ_NET_PROFILE (profiles/microbench/build_netprof.sh):

    python profiles/microbench/delta_resident_phases.py [games ...]     (default: 1, 256, 512 games; a whole 800-simulation search each)
"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import rlzero_amd._hip as H
H.library_path = lambda: os.environ.get('RZ_NETPROF_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librlzero_netprof.so')
import rlzero_amd._build as B
B.needs_build = lambda: False
import numpy as np, torch
from rlzero_amd.engine import HipNetEvaluator, MCTSEngine
from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
lib = H.load()
NAMES = [(0, 'leaf + changed cells'), (1, 'distances, ballots'), (2, 'bar'), (3, 'maps, base records'), (4, 'bar'), (5, 'conv1'), (6, 'bar'), (7, 'conv2'),
         (9, 'bar + conv3 + heads'), (10, 'bar'), (11, 'features'), (12, 'bar'), (16, 'value layer'), (17, 'expand + backup'), (18, 'selection')]
sizes = [int(a) for a in sys.argv[1:]] or [1, 256, 512]
for games in sizes:
    torch.manual_seed(0)
    net = PolicyValueNet(15).to('cuda:0')
    ev = HipNetEvaluator(net, 15, 'cuda:0', max_boards=games)
    sims = 800
    eng = MCTSEngine(15, 5, n_games=games, n_playout=sims, device='cuda:0', add_noise=True)
    eng.reset_games()
    assert ev.resident_ok(eng) and ev.resident_delta_ok(eng)
    # a position a few moves into a game (the windows of an empty board's first leaves touch its corner only)
    rng = np.random.default_rng(0)
    for ply in range(6):
        eng.sim_chunk(ev, 40)
        visits = eng.root_visits()
        moves = np.array([int(rng.choice(np.flatnonzero(v > 0))) for v in visits], dtype=np.int32)
        eng.advance(moves)
        eng.step(moves)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev.hip.delta_stats(reset=True)
    a.record(); eng.sim_chunk(ev, sims); b.record(); torch.cuda.synchronize()
    st = ev.hip.delta_stats()
    out = (ctypes.c_longlong * 24)()
    assert lib.rz_net_debug_profile(out) == 0
    us = 1e3 * a.elapsed_time(b) / sims
    total = out[23] / sims
    print('%d games, %d simulations: %.2f us per simulation of a game (events, bases included), %d cycles (%.2f GHz); %.2f conv3 / %.2f conv2 tiles, %.2f changed cells per leaf, %d leaves without a base' % (
        games, sims, us, total, total / us / 1e3, st['tiles3'] / max(1, st['delta'] + st['no_base']), st['tiles2'] / max(1, st['delta'] + st['no_base']),
        st['cells'] / max(1, st['delta'] + st['no_base']), st['no_base']))
    print('   ' + '  '.join('%s=%d' % (nm, out[k] / sims) for k, nm in NAMES), flush=True)
    eng.close(); ev.hip.close()
