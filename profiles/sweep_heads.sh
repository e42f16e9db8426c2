# Heads GEMM variants x lane layouts (run from the repository root on a GPU box).  Columns: run, sims/s, ms per move,
# trunk launch ms (union of the lanes' event intervals / launches), per-stream average, the trunk alone, roofline frac.
# l1 = 1 lane x 512 games, l2 = 2 lanes x 672 games (trunk capped at 224 workgroups).
mkdir -p gpurun_out/s5
echo '# run  sims/s  ms_per_move  trunk_ms(union of the lanes)  trunk_ms(per stream)  trunk_ms(alone)  roofline_frac   (l1 = 1 lane x 512 games, l2 = 2 lanes x 672 games, trunk capped at 224 workgroups)'
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-games-leg --no-literal-config "$@" > gpurun_out/s5/$name.json 2> gpurun_out/s5/$name.err; python - <<PY
import json
try:
    r=json.load(open('gpurun_out/s5/$name.json'))
    print('$name', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline'].get('avg_launch_ms_per_stream'), r['roofline'].get('exclusive_launch_ms'), r['roofline']['frac'])
except Exception as e:
    print('$name', 'FAILED', e)
PY
}
for h in f32 split32 split64; do
  run l1_$h --heads-algo $h --lanes 1 --games 512
done
for h in f32 split32 split64; do
  run l2_$h --heads-algo $h
done
run l2_split64_g896 --heads-algo split64 --games 896
run l2_split64_w232 --heads-algo split64 --trunk-wgs 232 --games 1392
