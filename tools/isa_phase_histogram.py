#!/usr/bin/env python3
"""Per-phase, per-class instruction histogram of one kernel from a `hipcc -gline-tables-only
-save-temps` .s file (round 5: "what are the non-arithmetic VALU instructions of the EM kernel").

    python tools/isa_phase_histogram.py file.s <mangled-kernel-name> [--blocks] [--weights w.json]

Every instruction carries the `.loc` of the source line it was inlined from; lines are mapped to
the phase functions of csrc/cacgmm_em.hpp by the line ranges of their definitions (found by
scanning the header for the `static __device__ ... phase_x(` lines), helper headers
(wave_la.hpp, pbbss_dev.hpp) are attributed to the phase of the nearest preceding cacgmm_em.hpp
line INSIDE the same basic block, else to the block's majority phase.  `--blocks` lists the basic
blocks (label, size, phase mix) so that loop trip counts can be written down by hand;
`--weights` takes {label: executions per wave-iteration} and prints the weighted table.
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'pb_bss_amd', 'csrc', 'cacgmm_em.hpp')

CLASSES = ['fma64', 'cvt', 'f64_other', 'xlane', 'valu_int', 'valu_mov', 'valu_other', 'lds',
           'vmem', 'salu', 'wait', 'branch']


def classify(op, line):
    if op.startswith('v_'):
        if 'permlane' in op or 'readlane' in op or 'writelane' in op or 'readfirstlane' in op:
            return 'xlane'
        if op.startswith('v_mov_b') and 'dpp' in line:
            return 'xlane'
        if op.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64', 'v_pk_fma',
                          'v_pk_mul', 'v_pk_add')):
            return 'fma64'      # (DPP operand forms of the FMA included: they carry a flop)
        if op.startswith('v_cvt_f64') or op.startswith('v_cvt_f32_f64'):
            return 'cvt'
        if '_f64' in op:
            return 'f64_other'  # rcp, frexp, ldexp, max/min, cmp, cndmask pairs, trig_preop ...
        if op.startswith(('v_mov_b', 'v_accvgpr', 'v_swap')):
            return 'valu_mov'
        if op.startswith(('v_add_u', 'v_add_co', 'v_addc', 'v_sub', 'v_mul_lo', 'v_mul_hi', 'v_mad_',
                          'v_lshl', 'v_lshr', 'v_ashr', 'v_and', 'v_or', 'v_xor', 'v_bfe', 'v_bfi',
                          'v_add3', 'v_lshl_add', 'v_add_lshl', 'v_cmp', 'v_cndmask', 'v_min_',
                          'v_max_', 'v_add_nc', 'v_not', 'v_mbcnt', 'v_med3', 'v_perm')):
            return 'valu_int'
        return 'valu_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_'):
        if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep')):
            return 'wait'
        if op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc', 's_endpgm')):
            return 'branch'
        return 'salu'
    return 'valu_other'


def phase_ranges():
    """[(first line, last line, name)] of the functions of struct EmKernel, by definition order."""
    names = []
    with open(HDR) as f:
        for n, l in enumerate(f, 1):
            m = re.match(r'\s+static __device__ (?:__forceinline__ )?[\w:<>\s\*&]+?\b(\w+)\(', l)
            if m:
                names.append((n, m.group(1)))
    out = []
    for (n, name), nxt in zip(names, names[1:] + [(10 ** 9, '')]):
        out.append((n, nxt[0] - 1, name))
    return out


PHASE_OF = {  # function -> phase column
    'phase_e': 'E', 'quad_forms': 'E', 'quad_forms_pipelined': 'E', 'load_frame': None,
    'phase_m': 'M', 'factor_class': 'F', 'cov_entry': 'F', 'store_apack': 'F',
    'inverse_from_eig': 'F', 'pair_index': 'F', 'prep_from_model': 'init',
    'phase_load': 'init', 'phase_init_gamma': 'init', 'mweight': None, 'run': 'glue',
    'carve': 'glue', 'carve_small_into': 'glue', 't_stride': None,
}


def main():
    path, kern = sys.argv[1], sys.argv[2]
    want_blocks = '--blocks' in sys.argv
    weights = None
    if '--weights' in sys.argv:
        weights = json.load(open(sys.argv[sys.argv.index('--weights') + 1]))
    ranges = phase_ranges()

    def func_of(line):
        for lo, hi, name in ranges:
            if lo <= line <= hi:
                return name
        return None

    files = {}
    started = False
    blocks = []            # (label, [(class, phase, op)])
    cur, label = [], 'entry'
    last_em_phase = None
    cur_file, cur_line = None, None
    inl = None
    with open(path) as f:
        for raw in f:
            s = raw.strip()
            m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
            if m:
                files[int(m.group(1))] = os.path.basename(m.group(2))
                continue
            if not started:
                if raw.startswith(kern + ':'):
                    started = True
                continue
            if s.startswith('.Lfunc_end'):
                break
            m = re.match(r'(\.LBB[0-9_]+):', s)
            if m:
                blocks.append((label, cur))
                cur, label = [], m.group(1)
                last_em_phase = None
                continue
            m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
            if m:
                cur_file, cur_line = files.get(int(m.group(1)), '?'), int(m.group(2))
                continue
            if not s or s.startswith((';', '.', '//')):
                continue
            op = s.split()[0]
            ph = None
            if cur_file == 'cacgmm_em.hpp':
                fn = func_of(cur_line)
                ph = PHASE_OF.get(fn, 'glue' if fn else None)
                if ph is None:
                    ph = last_em_phase
                else:
                    last_em_phase = ph
            else:
                ph = last_em_phase
            cur.append((classify(op, s), ph, op, cur_file, cur_line))
    blocks.append((label, cur))
    # unattributed instructions of a block take the block's majority phase
    table = collections.defaultdict(collections.Counter)
    for lab, ins in blocks:
        cnt = collections.Counter(p for _, p, *_ in ins if p)
        major = cnt.most_common(1)[0][0] if cnt else 'glue'
        w = 1.0
        if weights is not None:
            w = float(weights.get(lab, 0.0))
        mix = collections.Counter()
        for c, p, op, fl, ln in ins:
            p = p or major
            mix[p] += 1
            if w:
                table[p][c] += w
        if want_blocks and len(ins) >= 8:
            lines = [ln for *_, fl, ln in ins if fl == 'cacgmm_em.hpp']
            print(f'{lab:14s} n={len(ins):5d}  {dict(mix)}  em.hpp lines '
                  f'{min(lines) if lines else "-"}..{max(lines) if lines else "-"}'
                  + (f'  x{w:g}' if weights is not None else ''))
    if want_blocks and weights is None:
        return
    phases = [p for p in ('E', 'M', 'F', 'init', 'glue') if p in table]
    print('%-11s' % 'class' + ''.join('%10s' % p for p in phases) + '%10s' % 'all')
    for c in CLASSES:
        row = [table[p][c] for p in phases]
        print('%-11s' % c + ''.join('%10.0f' % v for v in row) + '%10.0f' % sum(row))
    valu = [sum(v for c, v in table[p].items() if c in ('fma64', 'cvt', 'f64_other', 'xlane',
                                                         'valu_int', 'valu_mov', 'valu_other'))
            for p in phases]
    print('%-11s' % 'VALU total' + ''.join('%10.0f' % v for v in valu) + '%10.0f' % sum(valu))


if __name__ == '__main__':
    main()
