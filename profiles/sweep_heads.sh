# heads GEMM variants x lane layouts (run from the repository root on a GPU box)
mkdir -p gpurun_out/s5
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-games-leg --no-literal-config "$@" > gpurun_out/s5/$name.json 2> gpurun_out/s5/$name.err; python - <<PY
import json
try:
    r=json.load(open('gpurun_out/s5/$name.json'))
    print('$name', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline'].get('exclusive_launch_ms'))
except Exception as e:
    print('$name', 'FAILED', e)
PY
}
for h in f32 split32 split64; do
  run l2_$h --heads-algo $h
  run l1_$h --heads-algo $h --lanes 1 --games 512
done
run l2_split64_g896 --heads-algo split64 --games 896
run l2_f32_g896 --heads-algo f32 --games 896
run l2_split64_w232 --heads-algo split64 --trunk-wgs 232 --games 1392
run l2_split64_w240 --heads-algo split64 --trunk-wgs 240 --games 1440
