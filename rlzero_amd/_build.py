"""Build recipe for the HIP extension (gfx950 only, in-tree, no JIT cache).

``python -m rlzero_amd._build`` or ``__graft_entry__.build()``.  hipcc cross-compiles
without a GPU.  -ffp-contract=off is REQUIRED: the tree arithmetic must round q + c*u
twice like CPython does.

The library carries the hash of what it was built from (sources, headers, flags): ``rz_source_hash()`` and the marker string
``RZ_SOURCE_HASH=<hex>`` in its bytes.  ``needs_build()`` compares that hash with the tree's -- not file times -- so a stale
binary is rebuilt wherever it came from, and ``_hip.load()`` refuses a library whose hash differs from the sources beside it.
Every source is compiled to an object of its own (in parallel; an object is reused while its own hash holds), then linked.
"""
import hashlib
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'librlzero_hip.so')
OBJ_DIR = os.path.join(PKG, 'csrc', '_obj')
SOURCES = [os.path.join(PKG, 'csrc', name) for name in ('rz_engine.hip', 'rz_net.hip', 'rz_muzero.hip')]
HEADERS = [os.path.join(REPO, "include", "rlzero_hip.h"), os.path.join(PKG, "csrc", "rz_trace.h"), os.path.join(PKG, "csrc", "rz_tree.h"), os.path.join(PKG, "csrc", "rz_delta.h")]
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fno-fast-math', '-fno-slp-vectorize', '-std=c++17',
         '-fPIC', '-Wall', '-Wno-unused-function']
MARKER = b'RZ_SOURCE_HASH='


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def source_hash():
    """Hash of everything the library is built from: the three sources, the headers, the flags."""
    return _digest(SOURCES + HEADERS, ' '.join(FLAGS))


def library_hash(path=LIB):
    """The hash a built library carries in its bytes, or None."""
    try:
        with open(path, 'rb') as f:
            m = re.search(MARKER + rb'([0-9a-f]{32})', f.read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def needs_build():
    return library_hash() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time (several ranks may call build() together): the others wait, then find the library current
    import fcntl
    with open(os.path.join(OBJ_DIR, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB
        return _build_locked(force, verbose)


def _build_locked(force, verbose):
    hipcc = os.environ.get('HIPCC', 'hipcc')
    want = source_hash()
    inc = ['-I' + os.path.join(REPO, 'include')]
    jobs, objs = [], []
    for src in SOURCES:
        # (rz_engine.hip holds rz_source_hash(): its object depends on the hash of the whole tree)
        own = _digest([src] + HEADERS, ' '.join(FLAGS) + (want if src.endswith('rz_engine.hip') else ''))
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + '.' + own + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj):
            for old in os.listdir(OBJ_DIR):
                if old.startswith(os.path.basename(src) + '.'):
                    os.remove(os.path.join(OBJ_DIR, old))
            # compiled beside its final name and renamed on success: a killed hipcc leaves no object to be reused
            tmp = '%s.tmp.%d' % (obj, os.getpid())
            cmd = [hipcc] + FLAGS + inc + ['-DRZ_SOURCE_HASH="%s"' % want, '-c', src, '-o', tmp]
            if verbose:
                print(' '.join(cmd), flush=True)
            jobs.append((subprocess.Popen(cmd), cmd, tmp, obj))
    failed = None
    for proc, cmd, tmp, obj in jobs:
        if failed is not None:
            proc.kill()
        rc = proc.wait()
        if rc == 0 and failed is None:
            os.replace(tmp, obj)
        else:
            if failed is None:
                failed = (rc, cmd)
            if os.path.exists(tmp):
                os.remove(tmp)
    if failed is not None:
        raise subprocess.CalledProcessError(*failed)
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(LIB + '.tmp', LIB)
    got = library_hash()
    if got != want:
        raise RuntimeError('the built library carries hash %r, the tree has %r' % (got, want))
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
