# Games per GPU (2 lanes, trunk on 224 CUs) and trunk workgroups x 4 boards (run from the repository root on a GPU box).
echo '# games_per_gpu  sims/s  ms_per_move  trunk_launch_ms(union of the lanes)  trunk_launch_ms(alone); last rows: trunk_workgroups games ...'
mkdir -p gpurun_out/s3
for g in 1344 1792 2240 2688; do
  python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --games $g > gpurun_out/s3/g$g.json 2> gpurun_out/s3/g$g.err
  python - <<PY
import json
r=json.load(open('gpurun_out/s3/g$g.json'))
print($g, r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['exclusive_launch_ms'])
PY
done
for w in 232 240; do
  g=$((w*2*4))
  python bench.py --no-cpu-baseline --no-games-leg --no-literal-config --games $g --trunk-wgs $w > gpurun_out/s3/w${w}_g$g.json 2> gpurun_out/s3/w${w}_g$g.err
  python - <<PY
import json
r=json.load(open('gpurun_out/s3/w${w}_g$g.json'))
print($w, $g, r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['exclusive_launch_ms'])
PY
done
