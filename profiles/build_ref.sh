#!/bin/bash
# build the HIP library of another revision (for same-box A / B runs through RZ_HIP_LIBRARY): profiles/build_ref.sh <rev> <out.so>
# Sources and headers come from <rev> (git archive), the compile flags from rlzero_amd/_build.py: the ref is built exactly as the
# in-tree library is.  RZ_HIP_LIBRARY swaps the library only: <rev> must have the ABI version of the working tree's binding
# (rlzero_amd/_hip.py) -- for older revisions run the ref leg from a `git worktree` of that revision instead.
set -e
rev=$1; out=$2; tmp=$(mktemp -d)
git archive $rev rlzero_amd/csrc include | tar -x -C $tmp
flags=$(python -c "from rlzero_amd import _build; print(' '.join(_build.FLAGS))")
srcs=$(python -c "from rlzero_amd import _build; import os; print(' '.join('$tmp/rlzero_amd/csrc/' + os.path.basename(s) for s in _build.SOURCES))")
hipcc $flags -I$tmp/include $srcs -o $out
rm -rf $tmp
