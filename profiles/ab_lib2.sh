# interleaved same-box A / B of the in-tree library against $REF, R rounds: M sims/s at $GAMES games
set -e
F="--no-cpu-baseline --no-fill --no-configs --no-games-leg --steps 8 --warmup 3 --regions 1 ${EXTRA}"
for rep in $(seq 1 ${R:-4}); do
for lib in ref new; do
  for games in ${GAMES:-512 1536}; do
    if [ $lib = ref ]; then export RZ_HIP_LIBRARY=$PWD/$REF; else unset RZ_HIP_LIBRARY; fi
    python bench.py $F --games $games | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib games $games:', round(d['value']/1e6, 3), d['ms_per_step'])"
  done
done
done
