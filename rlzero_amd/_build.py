"""Build recipe for the HIP extension (gfx950 only, in-tree, no JIT cache).

``python -m rlzero_amd._build`` or ``__graft_entry__.build()``.  hipcc cross-compiles
without a GPU.  -ffp-contract=off is REQUIRED: the tree arithmetic must round q + c*u
twice like CPython does.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'librlzero_hip.so')
SOURCES = [os.path.join(PKG, 'csrc', name) for name in ('rz_engine.hip', 'rz_net.hip', 'rz_muzero.hip')]
HEADERS = [os.path.join(REPO, "include", "rlzero_hip.h"), os.path.join(PKG, "csrc", "rz_trace.h"), os.path.join(PKG, "csrc", "rz_tree.h")]
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fno-fast-math', '-fno-slp-vectorize', '-std=c++17',
         '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


def needs_build():
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc] + FLAGS + ['-I' + os.path.join(REPO, 'include')] + SOURCES + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
