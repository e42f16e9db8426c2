#!/usr/bin/env python3
"""MFMA utilisation of the trunk from the PMC pass of profiles/pmc_mfma.sh / pmc_mfma_r03.sh -> <out>/keep/pmc_mfma.json.

    python3 profiles/summarise_mfma.py <out> [kernel name = k_trunk_split] [boards per launch = 512]

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md,
cycle constants); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 = the dispatch's cycles and,
divided by its duration from the kernel trace of the same command, the effective clock."""
import csv
import glob
import json
import os
import sys


def main(out, kernel='k_trunk_split', boards=512):
    keep = os.path.join(out, 'keep')
    os.makedirs(keep, exist_ok=True)
    found = glob.glob(os.path.join(out, 'pmc_mfma', '**', '*counter_collection.csv'), recursive=True)
    if not found:
        print('no counter file')
        return
    acc = {}
    for row in csv.DictReader(open(found[0])):
        if kernel not in row['Kernel_Name']:
            continue
        tot, n = acc.get(row['Counter_Name'], (0.0, 0))
        acc[row['Counter_Name']] = (tot + float(row['Counter_Value']), n + 1)
    mean = {k: tot / n for k, (tot, n) in acc.items() if n}
    launches = max([n for _, n in acc.values()] or [0])
    dur = []
    for f in glob.glob(os.path.join(out, 'pmc_mfma_trace', '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row['Kernel_Name']:
                dur.append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    dur_ns = sum(dur) / len(dur) if dur else None
    simds = 256 * 4
    rows = kernel == 'k_trunk_rows'
    # MFMAs per board and the busy cycles each counts: k_trunk_split 4 waves x 1080 of 32x32x16 (32 cycles); k_trunk_rows 4 waves x
    # (1548 + 387) of 16x16x32 (16 cycles) + conv1's 12 of 32x32x16 per wave (bitboard route)
    busy_expected = boards * 4 * ((1548 + 387) * 16 + 12 * 32) if rows else boards * 4 * 1080 * 32
    flops_per_busy_cycle = 16384.0 / 16.0 if rows else 32768.0 / 32.0
    rec = {'kernel': '%s, %d boards per launch, alone on the chip (1 lane, eager)' % (kernel, boards), 'launches': launches,
           'counters_mean_per_launch': {k: round(v, 1) for k, v in mean.items()}, 'avg_duration_ns_kernel_trace': dur_ns}
    mfma = mean.get('SQ_VALU_MFMA_BUSY_CYCLES')
    gui = mean.get('GRBM_GUI_ACTIVE')
    if mfma:
        rec['mfma_busy_cycles_expected'] = busy_expected
    if mfma and gui:
        cycles = gui / 8.0
        rec['dispatch_cycles'] = round(cycles, 1)
        rec['mfma_busy_fraction_of_simd_cycles'] = round(mfma / (cycles * simds), 4)
        if dur_ns:
            rec['effective_clock_ghz'] = round(cycles / dur_ns, 3)
            rec['executed_f16_mfma_tflops'] = round(mfma * flops_per_busy_cycle / dur_ns / 1e3, 1)
            rec['note'] = ('GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS '
                           'give-back), so dispatch_cycles is an upper bound of the SIMD cycles and the busy fraction a lower bound; '
                           'executed_f16_mfma_tflops = MFMA busy cycles x 1024 flops per cycle / duration, against the 2500 TFLOP/s peak')
    wc = mean.get('SQ_WAVE_CYCLES')
    if wc and mfma and dur_ns:
        waves = 256 * 4  # 256 persistent workgroups of 4 waves, one wave per SIMD, alive for the whole dispatch
        per_wave = 4.0 * wc / waves  # SQ_WAVE_CYCLES counts quad-cycles
        rec['wave_cycles_per_wave'] = round(per_wave, 1)
        rec['clock_ghz_from_wave_cycles'] = round(per_wave / dur_ns, 3)
        rec['mfma_busy_fraction_of_wave_cycles'] = round(mfma / simds / per_wave, 4)
    json.dump(rec, open(os.path.join(keep, 'pmc_mfma.json'), 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:3] or ['k_trunk_split']), *(int(a) for a in sys.argv[3:4]))
