#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks: one line per kernel (demangled name, VGPRs, AGPRs, SGPRs,
scratch bytes per lane, LDS bytes, occupancy).

    hipcc <flags of rlzero_amd/_build.py> -c rlzero_amd/csrc/rz_net.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> remarks.txt
    python tools/kernel_resources.py remarks.txt > profiles/rNN/kernel_resources_rz_net.txt
"""
import re
import subprocess
import sys


def main(path):
    rows, cur = [], None
    for line in open(path, errors='replace'):
        m = re.search(r'remark: .*?Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark: .*?\s{2,}([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = [r['name'] for r in rows]
    dem = names
    for tool in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', 'c++filt'):
        try:
            dem = subprocess.run([tool] + names, stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
            break
        except Exception:  # noqa: BLE001
            continue
    print('%-8s %-6s %-6s %-8s %-8s %-5s  %s' % ('VGPRs', 'AGPRs', 'SGPRs', 'scratch', 'LDS', 'occ', 'kernel'))
    for r, d in sorted(zip(rows, dem), key=lambda rd: rd[1]):
        d = re.sub(r'^void ', '', d).replace('(anonymous namespace)::', '')
        d = re.sub(r'\((?!anonymous).*$', '', d)   # (the argument list)
        print('%-8d %-6d %-6d %-8d %-8d %-5d  %s' % (r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('TotalSGPRs', -1), r.get('ScratchSize', -1),
                                                 r.get('LDS Size', -1), r.get('Occupancy', -1), d))


if __name__ == '__main__':
    main(sys.argv[1])
