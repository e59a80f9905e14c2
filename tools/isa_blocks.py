#!/usr/bin/env python3
"""Static per-basic-block instruction histogram of one kernel in a `hipcc -save-temps` .s file.

    python tools/isa_blocks.py file.s <kernel-name-substring> [min_block_size]

Prints, per basic block: instruction count split into FP64 VALU, other VALU, cross-lane
(DPP / permlane / readlane / writelane), SALU, LDS, VMEM/scratch.  Used to see where the EM
kernel's issue slots go (the kernel is VALU-issue bound).
"""
import re, sys, collections

def classify(op, line):
    if op.startswith('v_'):
        if 'dpp' in line or 'permlane' in op or 'readlane' in op or 'writelane' in op or 'readfirstlane' in op:
            return 'xlane'
        if '_f64' in op or op in ('v_frexp_mant_f64',):
            return 'f64'
        return 'valu'
    if op.startswith('s_'):
        if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'):
            return 'wait'
        if op.startswith('s_cbranch') or op.startswith('s_branch'):
            return 'br'
        if op.startswith('s_load') or op.startswith('s_buffer'):
            return 'smem'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        return 'scratch' if op.startswith('scratch_') else 'vmem'
    return 'other'

def main():
    path, kern = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[_A-Za-z0-9]+:', l) and kern in l and not l.startswith('.L'):
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    blocks, cur, name = [], collections.Counter(), lines[start].rstrip(':')
    tot = collections.Counter()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('s_endpgm'):
            blocks.append((name, cur))
            break
        m = re.match(r'^(\.LBB[0-9_]+):', s)
        if m:
            blocks.append((name, cur))
            cur, name = collections.Counter(), m.group(1)
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        c = classify(op, s)
        cur[c] += 1
        tot[c] += 1
    keys = ['f64', 'valu', 'xlane', 'salu', 'lds', 'vmem', 'scratch', 'smem', 'wait', 'br']
    print('%-14s %6s ' % ('block', 'total') + ' '.join('%7s' % k for k in keys))
    for name, c in blocks:
        n = sum(c.values())
        if n >= minsz:
            print('%-14s %6d ' % (name[-14:], n) + ' '.join('%7d' % c[k] for k in keys))
    print('%-14s %6d ' % ('ALL', sum(tot.values())) + ' '.join('%7d' % tot[k] for k in keys))

if __name__ == '__main__':
    main()
