#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh into the small files kept under
profiles/<round>/: per-kernel statistics (CSV as rocprofv3 wrote them) and pmc_traffic.json
(HBM bytes per launch of the three kernels of a simulation step; the FC GEMM is k_heads_split beside a capped trunk, k_heads_gemm otherwise).

    python3 profiles/summarise.py gpurun_out/r02

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md, section HBM: FETCH_SIZE and
WRITE_SIZE are collected in separate passes; their unit is KB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B / lane) streaming reads at 64 bytes, so reads are doubled (an upper
bound for kernels that also issue narrow reads); WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = {'k_trunk': 'k_trunk', 'k_tree_step': 'k_tree_step_raw', 'k_heads_gemm': 'k_heads_gemm', 'k_heads_split': 'k_heads_split', 'k_heads_part': 'k_heads_part'}


def per_kernel(csv_path, counter):
    acc = {}
    for row in csv.DictReader(open(csv_path)):
        if row['Counter_Name'] != counter:
            continue
        for key, needle in KERNELS.items():
            if needle in row['Kernel_Name']:
                tot, n = acc.get(key, (0.0, 0))
                acc[key] = (tot + float(row['Counter_Value']), n + 1)
    return {k: (tot / n, n) for k, (tot, n) in acc.items() if n}


def main(out):
    keep = os.path.join(out, 'keep')
    os.makedirs(keep, exist_ok=True)
    for name in ('bench_default.json', 'lane_sweeps.txt', 'in_flight_sweep.txt', 'bench_eager_literal_under_rocprof.json', 'bench_eager_c2_k16_under_rocprof.json', 'bench_muzero_under_rocprof.json',
                 'microbench_f32_mfma_overlap.txt', 'microbench_f16_mfma_rate.txt', 'microbench_f16_mfma_fillers.txt',
                 'bench_under_rocprof.json', 'sweep_games.txt', 'sweep_heads.txt',
                 'bench_eager_under_rocprof.json', 'bench_eager_1lane_under_rocprof.json',
                 'bench_c1_ttt.json', 'bench_c2_9x9.json', 'bench_c3_connect4.json', 'bench_c5_muzero_cartpole.json'):
        src = os.path.join(out, name)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(keep, name))
    for tag in ('stats_default', 'stats_eager', 'stats_eager_1lane', 'stats_eager_literal', 'stats_eager_c2_k16', 'stats_muzero'):
        found = glob.glob(os.path.join(out, tag, '**', '*kernel_stats.csv'), recursive=True)
        if found:
            shutil.copy(found[0], os.path.join(keep, 'bench_%s_kernel_stats.csv' % tag[6:]))
    traffic = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        found = glob.glob(os.path.join(out, 'pmc_' + counter, '**', '*counter_collection.csv'), recursive=True)
        if not found:
            continue
        shutil.copy(found[0], os.path.join(keep, 'pmc_%s_counter_collection.csv' % counter.lower()))
        for k, (mean_kb, n) in per_kernel(found[0], counter).items():
            traffic.setdefault(k, {})[counter] = (mean_kb, n)
    if traffic:
        line = {}
        try:
            line = json.loads(open(os.path.join(out, 'bench_default.json')).read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError):
            pass
        rec = {'workload': line.get('config', {}).get('workload'), 'lanes': line.get('config', {}).get('lanes'),
               'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of '
                         '`bench.py --no-cpu-baseline --steps 1 --warmup 0 --playouts 40 --graph 0`; per-dispatch '
                         'means; counter unit KB; reads doubled (gfx950 FETCH_SIZE counts 128-B requests of '
                         '16 B/lane reads at 64 B, MI355X_MICROARCH.md section HBM), writes exact',
               'kernels': {}}
        for k, c in traffic.items():
            fetch_kb, n = c.get('FETCH_SIZE', (0.0, 0))
            write_kb, _ = c.get('WRITE_SIZE', (0.0, 0))
            rec['kernels'][k] = {'launches': n, 'fetch_size_kb': round(fetch_kb, 1), 'write_size_kb': round(write_kb, 1),
                                 'traffic_bytes_per_launch': int(round((2.0 * fetch_kb + write_kb) * 1024))}
        json.dump(rec, open(os.path.join(keep, 'pmc_traffic.json'), 'w'), indent=1)
    print('kept:', sorted(os.listdir(keep)))


if __name__ == '__main__':
    main(sys.argv[1])
