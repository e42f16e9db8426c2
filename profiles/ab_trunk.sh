# same-box A / B of trunk builds through the bench (M sims/s, ms per move, us per trunk launch): ALGOS = --net-algo values
set -e
mkdir -p gpurun_out
F="--no-cpu-baseline --no-literal-config --no-configs --no-games-leg --steps 8 --warmup 3"
for rep in 1 2; do
for algo in ${ALGOS:-split_f16_tiles split_f16}; do
  for games in 0 512; do
    python bench.py $F --games $games --net-algo $algo | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$algo games $games:', round(d['value']/1e6, 3), d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))"
  done
done
done
