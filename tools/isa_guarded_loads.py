#!/usr/bin/env python3
"""Static check for the "guarded load" code-generation trap: compile the HIP
sources to gfx950 assembly and list, per kernel, how many global loads are DIRECTLY followed by
`s_waitcnt vmcnt(0)` -- each of those is a memory round trip that nothing overlaps.  Spin loops
on control words (sc1 loads of the inter-workgroup barriers) are expected hits; data loads in
streaming loops are not.

    python tools/isa_guarded_loads.py [--all]      # --all adds the D = 8 EM / Watson / joint units (minutes)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pb_bss_amd', 'csrc')
UNITS = ['embed', 'gauss_full', 'generic', 'generic_bf', 'beamform', 'bf_extra', 'dhtv', 'stft']
HEAVY = [('em_inst', 8), ('cw_inst', 8), ('joint_inst', 8)]


def asm(unit, d=None):
    out = tempfile.mktemp(suffix='.s')
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I.', '-I../../include',
           '-S', '--cuda-device-only', '-o', out, unit + '.hip']
    if d:
        cmd.insert(1, f'-DPBBSS_EM_D={d}')
    subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().split('\n')
    os.unlink(out)
    return text


def scan(lines):
    cur, stats = None, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            cur = m.group(1)
            stats[cur] = [0, 0, 0]
        t = l.strip()
        if cur and (t.startswith('global_load') or t.startswith('flat_load')):
            stats[cur][0] += 1
            j = i + 1
            while j < len(lines) and (not lines[j].strip() or lines[j].strip()[0] in ';.'):
                j += 1
            if j < len(lines) and 's_waitcnt vmcnt(0)' in lines[j]:
                stats[cur][1] += 1
                stats[cur][2] += ' sc1' in t  # control-word polls of the inter-workgroup barriers
    return stats


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names),
                             capture_output=True, text=True, check=True).stdout.split('\n')
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    units = [(u, None) for u in UNITS] + (HEAVY if '--all' in sys.argv else [])
    print('# unit | kernel | global loads | directly followed by s_waitcnt vmcnt(0) | of those sc1 (barrier polls)')
    for unit, d in units:
        st = scan(asm(unit, d))
        names = demangle(list(st))
        for k, (n, w, p) in sorted(st.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
            if w - p >= 3:
                print(f'{unit} | {names[k][:110]} | {n} | {w} | {p}')


if __name__ == '__main__':
    main()
