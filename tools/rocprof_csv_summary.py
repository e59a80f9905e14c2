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


def main(root, cmd, sha=''):
    print(f'# command: {cmd}')
    if sha:
        # bench.py only reads PMC figures from a summary whose hash matches the tree it runs in
        print(f'# kernel_source_sha: {sha}')
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
        # shader clock of the pass that carried GRBM_GUI_ACTIVE: cycles per XCD / kernel time
        if any(c == 'GRBM_GUI_ACTIVE' for (_, c) in acc):
            kt = glob.glob(os.path.join(os.path.dirname(f), '*kernel_trace.csv'))
            if kt:
                dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp'])
                       for r in csv.DictReader(open(kt[0])) if short(r['Kernel_Name']) == 'main']
                gui = acc.get(('main', 'GRBM_GUI_ACTIVE'))
                if dur and gui:
                    ghz = (sum(gui) / len(gui)) / 8.0 / (sum(dur) / len(dur))
                    print(f'# shader clock of the EM kernel in that pass: GRBM_GUI_ACTIVE / 8 XCDs / '
                          f'duration ({sum(dur) / len(dur) / 1e3:.1f} us)')
                    print(f'main_clock_ghz | {ghz:.3f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '', sys.argv[3] if len(sys.argv) > 3 else '')
