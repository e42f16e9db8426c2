#!/bin/bash
# the HBM-traffic passes of profiles/collect_r06.sh alone, merged into profiles/r06/pmc_traffic.json (RZ_PMC_MERGE)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r06p
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/r06"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0"
pmc() {
    tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- $B --graph 0 --steps 1 --warmup 1 "$@" > "$OUT/pmc_${tag}.json" 2> /dev/null
        echo "pmc $tag $c done"
    done
}
pmc default
if [ "${1:-}" != default ]; then RZ_RESIDENT=0 pmc lanes4; fi
cd "$ROOT" && RZ_PMC_MERGE="$ROOT/profiles/r06/pmc_traffic.json" python3 profiles/summarise_r06.py "$OUT" && cp "$OUT/keep/pmc_traffic.json" "$ROOT/gpurun_out/r06/pmc_traffic.json"
