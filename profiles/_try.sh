mkdir -p gpurun_out/s11
python -m pytest tests/test_gpu_parity.py -m gpu -x -q  > gpurun_out/s11/pytest.log 2>&1 ; tail -3 gpurun_out/s11/pytest.log
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-games-leg --no-literal-config "$@" > gpurun_out/s11/$name.json 2> gpurun_out/s11/$name.err; python - <<PY
import json
try:
    r=json.load(open('gpurun_out/s11/$name.json'))
    print('$name', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline'].get('exclusive_launch_ms'), r['roofline']['frac'])
except Exception as e:
    print('$name', 'FAILED', e)
PY
}
run l2
run l1 --lanes 1 --games 512
run l2b
run l1b --lanes 1 --games 512
