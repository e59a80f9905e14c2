#!/usr/bin/env python3
"""Summarise the csv output of tools/profile_round.sh (an un-profiled run, rocprofv3
--kernel-trace --stats and --pmc passes of the same command) into the text committed under
profiles/.  Only the MEASURED launches (the last `steps` of a pass; pre-heat and warm-up launches
are dropped) enter the duration statistics."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict


def short(name):
    if 'split_kernel' in name:
        return 'split'
    if 'cacgmm_em_kernel' in name:
        return 'main'
    if 'cacgmm_em32' in name:
        return 'main32'
    return name[:60]


def main(root, cmd, sha='', steps=25):
    steps = int(steps)
    print(f'# command: {cmd}')
    if sha:
        # bench.py only reads PMC figures from a summary whose hash matches the tree it runs in
        print(f'# kernel_source_sha: {sha}')
    print('# (F=513 T=500 D=8 K=3, 100 EM iterations + final E-step per step; each step = '
          'cacgmm_em_kernel (512 bins, the caller\'s stream) + cacgmm_em_split_kernel (bin 512 as 8 '
          'member workgroups, side stream) concurrently; the HIP events bracket both)')
    un = None
    try:
        with open(os.path.join(root, 'unprofiled.json')) as f:
            un = json.loads(f.read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f'# un-profiled run: not available ({type(e).__name__}: {e})')
    if un:
        print('# un-profiled run of the same command on the same box, right before the trace '
              '(HIP events inside the library around the launch; wall clock per step)')
        print(f"unprofiled_kernel_ms_hip_events | {un['roofline']['kernel_ms']:.4f}")
        print(f"unprofiled_ms_per_step | {un['ms_per_step']:.4f}")
        print(f"unprofiled_value_it_per_s | {un['value']:.1f}")
    med = None
    for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if short(r['Kernel_Name']) == 'main']
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
        meas = dur[-steps:]
        med = statistics.median(meas)
        print(f'# rocprofv3 --kernel-trace: {len(dur)} launches of the EM kernel in the pass '
              f'(pre-heat + warm-up + measured); statistics over the LAST {len(meas)} (= the '
              f'measured steps): em_kernel_trace_us | median | min | max | n')
        print(f'em_kernel_trace_us | {med:.1f} | {min(meas):.1f} | {max(meas):.1f} | {len(meas)}')
        print(f'# first five launches of the pass (clock ramp from idle): '
              f'{", ".join(f"{x:.0f}" for x in dur[:5])} us')
        if un:
            print(f"profiler_overhead_ratio_trace_median_over_hip_events | "
                  f"{med / (un['roofline']['kernel_ms'] * 1e3):.4f}")
        res = {k: rows[-1].get(k) for k in (
            'Grid_Size_X', 'Workgroup_Size_X', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count',
            'Accum_VGPR_Count', 'SGPR_Count')}
        print('# dispatch resources (EM kernel):', res)
        srows = [r for r in csv.DictReader(open(f)) if short(r['Kernel_Name']) == 'split']
        srows.sort(key=lambda r: int(r['Start_Timestamp']))
        sd = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in srows][-steps:]
        if sd:
            print('# concurrent split kernel of the remainder bin, same launches: '
                  'split_kernel_trace_us | median | min | max | n')
            print(f'split_kernel_trace_us | {statistics.median(sd):.1f} | {min(sd):.1f} | '
                  f'{max(sd):.1f} | {len(sd)}')
    for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_stats.csv'), recursive=True):
        print('# rocprofv3 --stats of the whole pass (ALL launches, pre-heat included): '
              'name | calls | avg_us | min_us | max_us | pct')
        for r in csv.DictReader(open(f)):
            print(f"# {r['Name'][:90]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                  f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']}")
    print('# PMC passes (separate runs, --kernel-trace --pmc <group>; measured launches only): '
          'kernel | counter | launches | mean per launch')
    for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'pbbss' in r['Kernel_Name']:
                acc[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            v = v[-steps:]
            print(f'{k} | {c} | {len(v)} | {sum(v)/len(v):.1f}')
        # shader clock of the pass that carried GRBM_GUI_ACTIVE: cycles per XCD / kernel time
        if any(c == 'GRBM_GUI_ACTIVE' for (_, c) in acc):
            kt = glob.glob(os.path.join(os.path.dirname(f), '*kernel_trace.csv'))
            if kt:
                dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp'])
                       for r in csv.DictReader(open(kt[0])) if short(r['Kernel_Name']) == 'main']
                dur = dur[-steps:]
                gui = acc.get(('main', 'GRBM_GUI_ACTIVE'))
                if dur and gui:
                    gui = gui[-steps:]
                    ghz = (sum(gui) / len(gui)) / 8.0 / (sum(dur) / len(dur))
                    print(f'# shader clock of the EM kernel in that pass: GRBM_GUI_ACTIVE / 8 XCDs / '
                          f'duration ({sum(dur) / len(dur) / 1e3:.1f} us; PMC passes serialise '
                          f'dispatches)')
                    print(f'main_clock_ghz | {ghz:.3f}')


if __name__ == '__main__':
    main(*sys.argv[1:5])
