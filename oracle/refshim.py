"""Import the real reference (fgnt/pb_bss) sub-packages WITHOUT running
``pb_bss/__init__.py`` (which needs paderbox & friends).

Only usable where /root/reference exists (the build container).  Used by
oracle/make_golden.py and by tests marked ``needs_reference`` to pin the NumPy
restatement; never at run time on the GPU box, never by the product package.
"""
import functools
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'pb_bss'))


def load():
    """Register a stub ``pb_bss`` package whose __path__ points at the
    reference tree.  The reference is read-only: no bytecode is written."""
    if not available():
        raise RuntimeError('reference tree not present at ' + REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if 'pb_bss' not in sys.modules:
        pkg = types.ModuleType('pb_bss')
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'pb_bss')]
        sys.modules['pb_bss'] = pkg
    if 'cached_property' not in sys.modules:
        cp = types.ModuleType('cached_property')
        cp.cached_property = functools.cached_property
        sys.modules['cached_property'] = cp
    return sys.modules['pb_bss']
