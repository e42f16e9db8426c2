"""Per-stage shader-clock cycles of k_mz_search (development aid; see profiles/microbench/README.md).  Needs a build of the
library with -DRZ_MZ_PROFILE next to this file: librlzero_prof.so."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import rlzero_amd._hip as H
H.library_path = lambda: os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get('MZ_LIB', 'librlzero_prof.so'))
import torch
from rlzero_amd.muzero import CartPoleBatch, MuZeroNet, MuZeroSelfPlay
lib = H.load()
for G, gpw in ((4096, 16),):
    torch.manual_seed(0)
    net = MuZeroNet().to('cuda').eval()
    sp = MuZeroSelfPlay(net, CartPoleBatch(G, torch.device('cuda'), seed=0), n_sims=50, seed=0, fused=True)
    sp.tree.set_search_shape(gpw)
    for _ in range(3):
        sp.play_move()
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.rz_mz_debug_profile(out)
    v = list(out)
    names = ['select', 'gather', 'bar0', 'S1', 'bar1', 'S2', 'bar2', 'S3', 'bar3', 'S4', 'bar4', 'heads', 'grow_backup', 'prologue', 'epilogue', 'depth']
    print('G=%d gpw=%d total cycles/sim=%.0f' % (G, gpw, sum(v[:15]) / 50.0))
    print('  ' + '  '.join('%s=%.0f' % (n, x / 50.0) for n, x in zip(names, v)))
    sp.close()
