set -e
F="--no-configs --no-cpu-baseline --no-fill --no-games-leg --steps 10 --warmup 3"
python bench.py $F > gpurun_out/d_base.json 2>/dev/null
RZ_DIAG_SKIP_FC=1 python bench.py $F > gpurun_out/d_skipfc.json 2>/dev/null
python bench.py $F --noise 0 > gpurun_out/d_nonoise.json 2>/dev/null
RZ_DIAG_SKIP_FC=1 python bench.py $F --noise 0 > gpurun_out/d_skipfc_nonoise.json 2>/dev/null
RZ_DIAG_SKIP_FC=1 python bench.py $F --noise 0 --lanes 2 > gpurun_out/d_skipfc_nonoise_2l.json 2>/dev/null
RZ_DIAG_SKIP_FC=1 python bench.py $F --noise 0 --lanes 3 > gpurun_out/d_skipfc_nonoise_3l.json 2>/dev/null
python bench.py $F > gpurun_out/d_base2.json 2>/dev/null
for f in d_base d_skipfc d_nonoise d_skipfc_nonoise d_skipfc_nonoise_2l d_skipfc_nonoise_3l d_base2; do python -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['regions_sims_per_sec'], d.get('small_kernels'), d['roofline']['avg_launch_ms_per_stream'])
"; done
