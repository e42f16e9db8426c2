# same box, alternating: the deferred-priors route against the three-launch step (512 games, four lanes; then 2 / 3 lanes, fill, C-legs)
F="--no-configs --no-cpu-baseline --no-fill --no-games-leg --steps 10 --warmup 3"
show() { python -c "
import json,sys
d=json.loads(open('gpurun_out/$1.json').read().strip().splitlines()[-1])
rf=d['roofline']
print('$1', d['value'], d['regions_sims_per_sec'], d.get('small_kernels'), 'trunk/stream', rf['avg_launch_ms_per_stream'], 'frac', rf['frac'], 'in flight', rf['launches_in_flight'])
"; }
for rep in 1 2; do
python bench.py $F --deferred 1 > gpurun_out/ab_def_$rep.json 2>/dev/null; show ab_def_$rep
python bench.py $F --deferred 0 > gpurun_out/ab_old_$rep.json 2>/dev/null; show ab_old_$rep
done
python bench.py $F --deferred 1 --lanes 2 > gpurun_out/ab_def_2l.json 2>/dev/null; show ab_def_2l
python bench.py $F --deferred 1 --lanes 3 > gpurun_out/ab_def_3l.json 2>/dev/null; show ab_def_3l
python bench.py $F --deferred 1 --games 1536 --lanes 2 > gpurun_out/ab_def_fill.json 2>/dev/null; show ab_def_fill
python bench.py $F --deferred 0 --games 1536 --lanes 2 > gpurun_out/ab_old_fill.json 2>/dev/null; show ab_old_fill
python bench.py $F --deferred 1 --games 768 --lanes 2 > gpurun_out/ab_def_768.json 2>/dev/null; show ab_def_768
python bench.py $F --deferred 1 --games 256 --lanes 1 > gpurun_out/ab_def_256.json 2>/dev/null; show ab_def_256
python bench.py $F --deferred 0 --games 256 --lanes 1 > gpurun_out/ab_old_256.json 2>/dev/null; show ab_old_256
