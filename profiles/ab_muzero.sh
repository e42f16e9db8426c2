# same-box A / B of two builds of the library on configs[4] (MuZero CartPole, 50 sims, 8192 envs): G sims/s
set -e
F="--game muzero --playouts 50 --games 8192 --steps 512 --warmup 48 --no-cpu-baseline --regions 1"
for rep in 1 2 3; do
for lib in ref new; do
    if [ $lib = ref ]; then export RZ_HIP_LIBRARY=$PWD/$REF; else unset RZ_HIP_LIBRARY; fi
    python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib:', round(d['value']/1e9, 4), d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_ms'))"
done
done
