#!/usr/bin/env python3
"""Summarise tools/profile_workload.sh (un-profiled run, rocprofv3 --kernel-trace --stats and
separate --pmc passes of `bench.py --workload config4|config5`) into the text committed under
profiles/.  A *step* is one fit (all EM iterations + final E-step); the number of steps in a pass
is the number of launches of the once-per-fit marker kernel, and every per-step figure is the
pass total of the library's kernels divided by that count (pre-heat, warm-up and measured steps
run the same launches)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

MARKER = {'config5': 'embed_prepare_kernel', 'config4': 'cwmm_em_',  # cwmm_em_kernel / cwmm_em_wide_kernel
          'config4_vmf': 'vmf_bin_em', 'config3': 'cacgmm_em_kernel'}        # vmf_bin_em_kernel / vmf_bin_em2_kernel
# config3's step is the whole chain of pipeline.separate: the torch copy / reduction kernels
# between the library's launches (transposes, the sum over the classes, the stack of the outputs)
# are part of the step and of its traffic, so they are listed and counted too
ALL_KERNELS = {'config3'}


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'(pbbss::[A-Za-z0-9_]+(<[^(]*>)?)', name)
    if m:
        return m.group(1)
    # torch kernels: the function name without its template arguments
    m = re.match(r'([A-Za-z0-9_:]+)', name)
    return 'torch: ' + (m.group(1) if m else name)[:60]


def rows_of(pattern):
    for f in glob.glob(pattern, recursive=True):
        yield from csv.DictReader(open(f))


def main(root, cmd, sha, workload):
    print(f'# command: {cmd}')
    print(f'# kernel_source_sha: {sha}')
    un = None
    try:
        with open(os.path.join(root, 'unprofiled.json')) as f:
            un = json.loads(f.read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f'# un-profiled run: not available ({type(e).__name__}: {e})')
    iters = 100
    if un:
        print('# un-profiled run of the same command on the same box, right before the trace')
        print(f"unprofiled_region_ms_hip_events | {un['roofline']['region_ms']:.4f}")
        print(f"unprofiled_ms_per_step | {un['ms_per_step']:.4f}")
        print(f"unprofiled_value_it_per_s | {un['value']:.1f}")
    # ---- kernel trace: per-kernel totals, steps from the marker kernel ----
    dur, calls = defaultdict(float), defaultdict(int)
    for r in rows_of(os.path.join(root, 'trace', '**', '*kernel_trace.csv')):
        if 'pbbss' not in r['Kernel_Name'] and workload not in ALL_KERNELS:
            continue
        k = short(r['Kernel_Name'])
        dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        calls[k] += 1
    marker = MARKER.get(workload)
    if marker:
        steps = sum(c for k, c in calls.items() if marker in k and 'split' not in k)
    else:  # vMF leg: (iterations + 1) launches of vmf_em_kernel per fit
        steps = sum(c for k, c in calls.items() if 'vmf_em_kernel' in k) // (iters + 1)
    steps = max(steps, 1)
    print(f'# rocprofv3 --kernel-trace: {steps} steps in the pass (pre-heat + warm-up + measured, all '
          f'alike); per kernel of the library: name | launches per step | avg us | us per step')
    tot = 0.0
    for k in sorted(dur, key=lambda k: -dur[k]):
        print(f'{k} | {calls[k] / steps:.2f} | {dur[k] / calls[k]:.2f} | {dur[k] / steps:.1f}')
        tot += dur[k]
    # kernels of the side stream (split kernels) run beside the main one: not on the critical path
    side = sum(v for k, v in dur.items() if 'split_kernel' in k)
    print('# sum of the kernel durations of one step (side-stream split kernels excluded: they run '
          'beside the main kernel):')
    print(f'region_trace_us | {(tot - side) / steps:.1f}')
    if un:
        print(f"trace_over_hip_events | {(tot - side) / steps / (un['roofline']['region_ms'] * 1e3):.4f}")
    # ---- PMC passes ----
    print('# PMC passes (separate runs, --kernel-trace --pmc <group>): kernel | counter | launches | '
          'mean per launch | total per step')
    step_tot = defaultdict(float)
    for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'),
                              recursive=True)):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'pbbss' in r['Kernel_Name'] or workload in ALL_KERNELS:
                acc[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
        if not acc:
            continue
        # steps of THIS pass (the profiler slows the pre-heat loop down: every pass runs its own
        # number of steps): launches of the once-per-fit marker kernel under one counter
        c0 = sorted({c for (_, c) in acc})[0]
        if marker:
            psteps = sum(len(v) for (k, c), v in acc.items()
                         if c == c0 and marker in k and 'split' not in k)
        else:
            psteps = sum(len(v) for (k, c), v in acc.items()
                         if c == c0 and 'vmf_em_kernel' in k) // (iters + 1)
        psteps = max(psteps, 1)
        print(f'# pass {[d for d in f.split(os.sep) if d.startswith("pmc_")][0]}: {psteps} steps')
        for (k, c), v in sorted(acc.items()):
            print(f'{k} | {c} | {len(v)} | {sum(v) / len(v):.1f} | {sum(v) / psteps:.1f}')
            step_tot[c] += sum(v) / psteps
    if 'FETCH_SIZE' in step_tot:
        print('# FETCH_SIZE / WRITE_SIZE are KiB at the L2 <-> fabric interface (Infinity-Cache hits '
              'included); raw counters, no gfx950 correction applied (MI355X_MICROARCH.md: FETCH_SIZE '
              'reports half the bytes of 16 B/lane streaming reads -- the corrected read figure is '
              'step_fetch_bytes_x2)')
        print(f"step_fetch_bytes | {step_tot['FETCH_SIZE'] * 1024:.0f}")
        print(f"step_fetch_bytes_x2 | {step_tot['FETCH_SIZE'] * 2048:.0f}")
    if 'WRITE_SIZE' in step_tot:
        print(f"step_write_bytes | {step_tot['WRITE_SIZE'] * 1024:.0f}")
    return step_tot
    if 'TCC_HIT_sum' in step_tot:
        h, m = step_tot['TCC_HIT_sum'], step_tot.get('TCC_MISS_sum', 0.0)
        print(f'l2_hit_rate | {h / max(h + m, 1.0):.4f}')


if __name__ == '__main__':
    main(*sys.argv[1:5])
