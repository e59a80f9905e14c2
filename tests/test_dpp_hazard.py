"""The DPP operand path of the float64 dot products (csrc/pbbss_dev.hpp: fmac_row_bcast) is inline
assembly, outside the compiler's hazard recogniser: tools/check_dpp_hazard.py proves on the
shipped binary that no VALU write of a DPP source register sits in the two issue slots in front
of the DPP instruction.  CPU-only (disassembly of the in-tree library)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location(
        'check_dpp_hazard', os.path.join(ROOT, 'tools', 'check_dpp_hazard.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scanner_flags_a_valu_write_in_front_of_a_dpp_read():
    t = _tool()
    dpp = 'v_fmac_f64_dpp v[0:1], v[10:11], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf'
    clean = f'''
<k>:
\tds_read_b64 v[10:11], v7
\ts_waitcnt lgkmcnt(0)
\t{dpp}
\tv_mul_f64 v[10:11], v[2:3], v[2:3]
\tv_add_f64 v[20:21], v[2:3], v[2:3]
\tv_add_f64 v[22:23], v[2:3], v[2:3]
\t{dpp}
\tv_mov_b32_e32 v11, v3
\ts_nop 1
\t{dpp}
'''
    n, bad = t.scan(clean)
    assert n == 3 and bad == []
    for prefix in ('\tv_mul_f64 v[10:11], v[2:3], v[2:3]\n',
                   '\tv_mov_b32_e32 v11, v3\n\tv_add_f64 v[20:21], v[2:3], v[2:3]\n',
                   '\tv_mov_b32_e32 v10, v3\n\ts_nop 0\n'):
        n, bad = t.scan(f'<k>:\n{prefix}\t{dpp}\n')
        assert n == 1 and len(bad) == 1, prefix


def test_shipped_library_has_no_dpp_hazard():
    t = _tool()
    lib = os.path.join(ROOT, 'pb_bss_amd', 'libpbbss_hip.so')
    if not os.path.exists(lib) or not os.path.exists(t.OBJDUMP):
        pytest.skip('library or llvm-objdump not available')
    import subprocess
    import tempfile
    total, violations = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for i, co in t.code_objects(lib):
            p = os.path.join(tmp, f'co{i}.o')
            with open(p, 'wb') as f:
                f.write(co)
            out = subprocess.run([t.OBJDUMP, '-d', '--no-show-raw-insn', p], capture_output=True,
                                 text=True, check=True).stdout
            n, bad = t.scan(out)
            total += n
            violations += bad
    assert total > 0, 'no DPP instruction found: did the extraction work?'
    assert violations == [], violations[:5]
