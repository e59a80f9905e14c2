#!/usr/bin/env python3
"""Summarise the csv output of tools/profile_round.sh (rocprofv3 --kernel-trace --stats and
--pmc passes) into the text committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if 'split_kernel' in name:
        return 'split'
    if 'cacgmm_em_kernel' in name:
        return 'main'
    return name[:60]


def main(root, cmd):
    print(f'# command: {cmd}')
    print('# (F=513 T=500 D=8 K=3, 100 EM iterations + final E-step per step; each step = '
          'cacgmm_em_kernel (512 bins) + cacgmm_em_split_kernel (bin 512, 8 workgroups) concurrently)')
    for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_stats.csv'), recursive=True):
        print('# rocprofv3 --kernel-trace --stats: name | calls | avg_us | min_us | max_us | pct')
        for r in csv.DictReader(open(f)):
            print(f"{r['Name'][:90]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                  f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']}")
    for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
        res = {}
        for r in csv.DictReader(open(f)):
            if 'pbbss' in r['Kernel_Name']:
                res[short(r['Kernel_Name'])] = {k: r.get(k) for k in (
                    'Grid_Size_X', 'Workgroup_Size_X', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count',
                    'Accum_VGPR_Count', 'SGPR_Count')}
        print('# dispatch resources:', res)
    print('# PMC passes (separate runs, --kernel-trace --pmc <group>): kernel | counter | launches | mean per launch')
    for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'pbbss' in r['Kernel_Name']:
                acc[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            print(f'{k} | {c} | {len(v)} | {sum(v)/len(v):.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
