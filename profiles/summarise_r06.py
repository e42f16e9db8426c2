#!/usr/bin/env python3
"""profiles/summarise_r03.py with round 6's kernels: k_delta_res (the resident search with the receptive-field trunk) and
k_trunk_delta (its two-launch step) get entries of their own -- a kernel name is credited to the LONGEST needle it contains.

    python3 profiles/summarise_r06.py /tmp/rz_r06
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarise_r03 as base   # noqa: E402

base.KERNELS = dict(base.KERNELS, k_delta_res='k_delta_res', k_trunk_delta='k_trunk_delta', k_trunk_rows='k_trunk_rows')
_plain = base.per_kernel


def per_kernel(csv_path, counter):
    import csv
    acc = {}
    for row in csv.DictReader(open(csv_path)):
        if row['Counter_Name'] != counter:
            continue
        hits = [k for k, needle in base.KERNELS.items() if needle in row['Kernel_Name']]
        if not hits:
            continue
        key = max(hits, key=lambda k: len(base.KERNELS[k]))
        tot, n = acc.get(key, (0.0, 0))
        acc[key] = (tot + float(row['Counter_Value']), n + 1)
    return {k: (tot / n, n) for k, (tot, n) in acc.items() if n}


base.per_kernel = per_kernel
if __name__ == '__main__':
    base.main(sys.argv[1])
