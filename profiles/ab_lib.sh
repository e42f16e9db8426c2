# same-box A / B of two builds of the library (the in-tree one against $REF) through the bench:
# M sims/s, ms per move, us per trunk launch; default workload and 512 games
set -e
mkdir -p gpurun_out
F="--no-cpu-baseline --no-literal-config --no-configs --no-games-leg --steps 8 --warmup 3 ${EXTRA}"
for rep in 1 2; do
for lib in ref new; do
  for games in ${GAMES:-512 1536}; do
    if [ $lib = ref ]; then export RZ_HIP_LIBRARY=$PWD/$REF; else unset RZ_HIP_LIBRARY; fi
    python bench.py $F --games $games | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib games $games:', round(d['value']/1e6, 3), d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))"
  done
done
done
