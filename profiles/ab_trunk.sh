set -e
mkdir -p gpurun_out
F="--no-cpu-baseline --no-literal-config --no-configs --no-games-leg --steps 8 --warmup 3"
for rep in 1 2; do
for algo in split_f16_tiles split_f16; do
  echo "== default $algo"; python bench.py $F --net-algo $algo | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))"
  echo "== 512 games $algo"; python bench.py $F --games 512 --net-algo $algo | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_launch_ms'))"
done
done
