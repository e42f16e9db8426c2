#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect_r03.sh into the small files kept under profiles/r03/: per-kernel
statistics (CSV as rocprofv3 wrote them) and pmc_traffic.json = {workload: {kernels: {kernel: HBM bytes per launch}}}
for every workload bench.py reports (bench.pmc_traffic reads it).

    python3 profiles/summarise_r03.py gpurun_out/r03

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md, section HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate passes; their unit is KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B / lane) streaming reads
at 64 bytes, so reads are doubled (an upper bound for kernels that also issue narrow reads); WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = {'k_trunk': 'k_trunk', 'k_tree_step': 'k_tree_step', 'k_heads': 'k_heads', 'k_mz_search': 'k_mz_search',
           'k_deferred_priors': 'k_deferred_priors', 'k_play_draw': 'k_play_draw', 'k_play_apply': 'k_play_apply'}


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
    for name in sorted(os.listdir(out)):
        if name.startswith('bench_') and name.endswith('.json') and os.path.getsize(os.path.join(out, name)):
            shutil.copy(os.path.join(out, name), os.path.join(keep, name))
    for d in sorted(glob.glob(os.path.join(out, 'stats_*'))):
        found = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)
        if found:
            shutil.copy(found[0], os.path.join(keep, 'bench_%s_kernel_stats.csv' % os.path.basename(d)[6:]))
    # how the trunk dispatches of the lanes overlap in the headline run (hipGraph replays): with four lanes two dispatches of 128
    # boards share the CUs, so the chip-level rate is boards / (wall covered by >= 1 dispatch), not boards / a dispatch's duration
    for tag in ('default', 'fill'):
        found = glob.glob(os.path.join(out, 'stats_%s' % tag, '**', '*kernel_trace.csv'), recursive=True)
        if not found:
            continue
        spans = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(found[0]))
                       if 'k_trunk' in r['Kernel_Name'])
        if len(spans) < 100:
            continue
        spans = spans[len(spans) // 4:]   # (skip the warm-up)
        covered, cur_a, cur_b = 0, spans[0][0], spans[0][1]
        for a, b in spans[1:]:
            if a > cur_b:
                covered += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        covered += cur_b - cur_a
        total = sum(b - a for a, b in spans)
        json.dump({'kernel': 'k_trunk_*', 'dispatches': len(spans), 'mean_dispatch_us': round(total / len(spans) / 1e3, 3),
                   'covered_by_a_dispatch_us_per_dispatch': round(covered / len(spans) / 1e3, 3),
                   'dispatches_in_flight': round(total / covered, 3),
                   'span_us_per_dispatch': round((spans[-1][1] - spans[0][0]) / len(spans) / 1e3, 3)},
                  open(os.path.join(keep, 'trunk_overlap_%s.json' % tag), 'w'), indent=1)
    traffic = {}
    for line_file in sorted(glob.glob(os.path.join(out, 'pmc_*.json'))):
        tag = os.path.basename(line_file)[4:-5]
        try:
            line = json.loads(open(line_file).read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError):
            continue
        workload = line['config']['workload']
        workload += {'puct': '+puct', 'c2k16': '+k16', '3launch': '+3launch', 'fp8': '+fp8', 'lanes4': '+lanes4'}.get(tag, '')   # same geometry, another rule / mode: its own entry
        rec = {'tag': tag, 'kernels': {}}
        per = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            found = glob.glob(os.path.join(out, 'pmc_%s_%s' % (tag, counter), '**', '*counter_collection.csv'), recursive=True)
            if found:
                for k, (mean_kb, n) in per_kernel(found[0], counter).items():
                    per.setdefault(k, {})[counter] = (mean_kb, n)
        for k, c in per.items():
            fetch_kb, n = c.get('FETCH_SIZE', (0.0, 0))
            write_kb, _ = c.get('WRITE_SIZE', (0.0, 0))
            rec['kernels'][k] = {'launches': n, 'fetch_size_kb': round(fetch_kb, 1), 'write_size_kb': round(write_kb, 1),
                                 'traffic_bytes_per_launch': int(round((2.0 * fetch_kb + write_kb) * 1024))}
        if rec['kernels']:
            traffic[workload] = rec
    merge = os.environ.get('RZ_PMC_MERGE')   # a pmc_traffic.json to update with the passes of this run (a partial re-collection)
    if traffic and merge and os.path.exists(merge):
        base = json.load(open(merge))
        base.update(traffic)
        traffic = base
    if traffic:
        traffic['_method'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `bench.py <the workload\'s flags> '
                              '--graph 0 --steps 1 --warmup 1` (eager launches, the workload\'s own playout count); per-dispatch means; '
                              'counter unit KB; reads doubled (gfx950 FETCH_SIZE counts 128-B requests of 16 B/lane reads at 64 B, '
                              'MI355X_MICROARCH.md section HBM), writes exact')
        json.dump(traffic, open(os.path.join(keep, 'pmc_traffic.json'), 'w'), indent=1)
    print('kept:', sorted(os.listdir(keep)))


if __name__ == '__main__':
    main(sys.argv[1])
