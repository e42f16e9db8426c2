# lanes x games under the deferred-priors route (15x15, 800 simulations per move): M sims/s, same box
F="--no-configs --no-cpu-baseline --no-fill --no-games-leg --steps 8 --warmup 3 --timeline 0 --regions 1"
for games in 128 192 256 320 384 448 512 640 768 1024 1536; do
  for lanes in 1 2 3 4; do
    if [ $((games / lanes)) -ge 32 ]; then
      python bench.py $F --games $games --lanes $lanes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('games $games lanes $lanes:', round(d['value']/1e6, 3))"
    fi
  done
done
