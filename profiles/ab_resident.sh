F="--no-configs --no-cpu-baseline --no-fill --no-games-leg --steps 8 --warmup 3 --timeline 0 --regions 1"
for games in 1 16 64 128 192 256; do
 for res in 1 0; do
  RZ_RESIDENT=$res python bench.py $F --games $games --lanes 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('15x15 games $games resident $res:', round(d['value']/1e6, 3))"
 done
done
RZ_RESIDENT=0 python bench.py $F --games 256 --lanes 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('15x15 games 256 two lanes:', round(d['value']/1e6, 3))"
