#!/usr/bin/env python3
"""Scan the device code of a built library for the DPP read-after-VALU-write hazard.

`fmac_row_bcast` (csrc/pbbss_dev.hpp) emits `v_fmac_f64_dpp ... row_newbcast:N` through inline
assembly, which the compiler's hazard recogniser does not look into: a VALU instruction that
writes the DPP source register must be followed by two wait states before the DPP instruction
reads it.  The operand registers are filled by LDS loads, so the sequence should never occur --
this tool proves it on the shipped binary: it unpacks the gfx950 code objects out of the
`.hip_fatbin` section (clang offload bundles), disassembles them with llvm-objdump and checks the
two issue slots in front of every DPP instruction (`s_nop N` counts N + 1 wait states).

    python tools/check_dpp_hazard.py [pb_bss_amd/libpbbss_hip.so]      exit 1 on a violation
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path):
    """Yield (index, bytes) of every amdgcn code object bundled into `path`."""
    blob = open(path, 'rb').read()
    pos, n = 0, 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        count, = struct.unpack_from('<Q', blob, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from('<QQQ', blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if 'amdgcn' in triple and size:
                yield n, blob[pos + off:pos + off + size]
                n += 1
        pos += len(MAGIC)


REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs(tok):
    m = REG.fullmatch(tok.strip().lstrip('-').strip('|'))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def scan(disasm):
    """-> (number of DPP instructions, list of violations) for one disassembly."""
    window = []  # (wait states this instruction provides, set of VGPRs it writes as a VALU op, text)
    n_dpp, bad = 0, []
    func = '?'
    for line in disasm.splitlines():
        line = line.split('//')[0].rstrip()
        if line.endswith('>:'):
            func = line.split('<')[-1][:-2]
            window = []
            continue
        text = line.strip()
        if not text or text.startswith(('.', ';')) or ':' in text.split()[0]:
            continue
        op, _, rest = text.partition(' ')
        ops = [x.strip() for x in rest.split(',')] if rest else []
        if '_dpp' in op:
            n_dpp += 1
            src = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            need = 2
            for states, writes, t in reversed(window):
                if need <= 0:
                    break
                if writes & src:
                    bad.append((func, t, text))
                    break
                need -= states
        states = 1
        writes = set()
        if op == 's_nop':
            states = int(ops[0], 0) + 1
        elif op.startswith('v_') and ops and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
            writes = regs(ops[0])
        window.append((states, writes, text))
        window = window[-4:]
    return n_dpp, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pb_bss_amd', 'libpbbss_hip.so')
    total, violations, nobj = 0, [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, co in code_objects(lib):
            if b'\xfa' not in co:  # DPP encodings carry src0 = 0xfa; cheap pre-filter
                continue
            p = os.path.join(tmp, f'co{i}.o')
            open(p, 'wb').write(co)
            out = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', p], capture_output=True,
                                 text=True, check=True).stdout
            n, bad = scan(out)
            total += n
            violations += bad
            nobj += 1
    print(f'{lib}: {nobj} code objects, {total} DPP instructions, {len(violations)} hazard violations')
    for func, writer, dpp in violations[:20]:
        print(f'  {func}: "{writer}"  ->  "{dpp}"')
    return 1 if violations else 0


if __name__ == '__main__':
    sys.exit(main())
