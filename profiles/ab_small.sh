# same-box A / B of two builds of the library on the small-batch configurations: sims/s, ms per move
set -e
run() { python bench.py "$@" --no-cpu-baseline --no-games-leg --no-fill --no-configs --regions 1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])"; }
for rep in 1 2; do
for lib in ref new; do
    if [ $lib = ref ]; then export RZ_HIP_LIBRARY=$PWD/$REF; else unset RZ_HIP_LIBRARY; fi
    echo "$lib C1 1 game:   $(run --board 3 --playouts 25 --games 1 --lanes 1 --steps 9 --warmup 20)"
    echo "$lib C1 16 games: $(run --board 3 --playouts 25 --games 16 --lanes 1 --steps 9 --warmup 20)"
    echo "$lib C2 64 games: $(run --board 9 --playouts 200 --games 64 --lanes 1 --steps 8 --warmup 8)"
    echo "$lib C2 256 games: $(run --board 9 --playouts 200 --games 256 --lanes 1 --steps 8 --warmup 8)"
done
done
