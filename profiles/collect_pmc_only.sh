#!/bin/bash
# A partial re-collection: the HBM counter passes of the given tags only (profiles/collect_r04.sh has the list), merged into
# profiles/r04/pmc_traffic.json ->  gpurun_out/r04/pmc_traffic.json.   bash profiles/collect_pmc_only.sh default fp8
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/rz_r04_pmc
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/r04"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 --timeline 0"
pmc() {
    tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- $B --graph 0 --steps 1 --warmup 1 "$@" > "$OUT/pmc_${tag}.json" 2> /dev/null
    done
}
for tag in "$@"; do
    case $tag in
        default) pmc default ;;
        fp8) pmc fp8 --net-algo split_f16_fp8 ;;
        fill) pmc fill --games 1536 ;;
        3launch) pmc 3launch --deferred 0 ;;
        *) echo "unknown tag $tag"; exit 1 ;;
    esac
done
cd "$ROOT" && RZ_PMC_MERGE="$ROOT/profiles/r04/pmc_traffic.json" python3 profiles/summarise_r03.py "$OUT" && cp "$OUT"/keep/pmc_traffic.json "$ROOT/gpurun_out/r04/"
