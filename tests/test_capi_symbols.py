"""CPU: the C-ABI library builds, loads and exports every symbol that
include/pbbss.h declares; status/enum constants of the ctypes layer match the
header.  No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'pbbss.h')


def header_text():
    with open(HEADER) as f:
        return f.read()


def declared_functions():
    txt = re.sub(r'/\*.*?\*/', '', header_text(), flags=re.S)
    return sorted(set(re.findall(r'\b(pbbss_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from pb_bss_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 18, names
    assert sorted(_lib.EXPORTS) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.pbbss_version() == 610
    assert lib.pbbss_error_string(0) == b'ok'
    assert b'shape' in lib.pbbss_error_string(-2)


def test_constants_match_header():
    from pb_bss_amd import _lib
    txt = header_text()

    def define(name):
        m = re.search(r'#define\s+' + name + r'\s+\(?(-?\d+)u?\)?', txt)
        assert m, name
        return int(m.group(1))

    assert define('PBBSS_ERR_INVALID_ARG') == _lib.ERR_INVALID_ARG
    assert define('PBBSS_ERR_UNSUPPORTED') == _lib.ERR_UNSUPPORTED
    assert define('PBBSS_ERR_HIP') == _lib.ERR_HIP
    assert define('PBBSS_ERR_LDS_CAPACITY') == _lib.ERR_LDS_CAPACITY
    assert define('PBBSS_ERR_INTERNAL') == _lib.ERR_INTERNAL
    # every error code of the header has its own text
    codes = [int(c) for c in re.findall(r'#define\s+PBBSS_ERR_[A-Z_]+\s+\((-\d+)\)', txt)]
    assert sorted(codes) == [-5, -4, -3, -2, -1]
    texts = {_lib.load().pbbss_error_string(c) for c in codes}
    assert len(texts) == len(codes) and b'unknown error' not in texts
    for n in ['NONFINITE', 'EIG_NOCONV', 'FLOORED', 'SLOWPATH', 'NOT_POSDEF', 'SINGULAR']:
        assert define('PBBSS_ST_' + n) == getattr(_lib, 'ST_' + n)
    assert define('PBBSS_COVNORM_EIGENVALUE') == _lib.COVNORM['eigenvalue']
    assert define('PBBSS_COVNORM_TRACE') == _lib.COVNORM['trace']
    assert define('PBBSS_COVNORM_NONE') == _lib.COVNORM[False]
    assert define('PBBSS_LAYOUT_TD') == _lib.LAYOUT_TD and define('PBBSS_LAYOUT_DT') == _lib.LAYOUT_DT
    assert define('PBBSS_WEIGHT_UNIFORM') == _lib.WEIGHT_UNIFORM


def test_em_opts_struct_layout():
    import ctypes
    from pb_bss_amd import _lib
    # 8 int32 + 2 double + 2 int32 (precision, reserved), naturally aligned: 56 bytes like the
    # C struct
    assert ctypes.sizeof(_lib.EmOpts) == 56
    assert _lib.EmOpts.affiliation_eps.offset == 32
    assert _lib.EmOpts.precision.offset == 48
    # pbbss_mix_opts: 8 int32 + 6 double + 2 int32 (sharded, reserved)
    assert ctypes.sizeof(_lib.MixOpts) == 88
    assert _lib.MixOpts.sharded.offset == 80


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when there is no GPU."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pb_bss_amd import _lib
    from pb_bss_amd.distribution import CACGMMTrainer
    y = np.zeros((2, 10, 3), np.complex64)
    with pytest.raises(_lib.PbbssError):
        CACGMMTrainer().fit(y, num_classes=2, iterations=1)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'pb_bss_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                with open(os.path.join(dirpath, f)) as fh:
                    src = fh.read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert '/root/reference' not in src, f


def test_shard_bounds_c_abi_matches_python():
    """pbbss_shard_bounds (pure host code, no GPU needed) == pb_bss_amd.sharding.shard_bounds."""
    import ctypes
    from pb_bss_amd import _lib
    from pb_bss_amd.sharding import shard_bounds
    lib = _lib.load()
    for F in (0, 1, 5, 8, 129, 257, 513):
        for world in (1, 2, 3, 8, 16):
            for rank in range(world):
                lo, hi = ctypes.c_int64(-1), ctypes.c_int64(-1)
                assert lib.pbbss_shard_bounds(F, world, rank, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_bounds(F, world, rank), (F, world, rank)
    bad = ctypes.c_int64(0)
    assert lib.pbbss_shard_bounds(5, 2, 2, ctypes.byref(bad), ctypes.byref(bad)) != 0
