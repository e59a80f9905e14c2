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


# ---------------------------------------------------------------------------------------------
# The reference's two native files (extraction/cythonized/get_gev_vector.pyx, c_eig.pyx) compiled
# out of tree: the sources are read where they lie under /root/reference, every output (generated
# C, objects, extension modules) goes to oracle/_ref/ (git-ignored, travels with gpurun).
REF_BUILD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')
_CYTHON_MODULES = ('get_gev_vector', 'c_eig')


def build_cython(force=False):
    """cythonize + compile both .pyx into oracle/_ref/.  Returns the list of built files.
    Needs the reference tree, Cython, a C compiler and SciPy's cython_lapack .pxd."""
    if not available():
        raise RuntimeError('reference tree not present at ' + REFERENCE_ROOT)
    import subprocess
    import sysconfig
    import numpy
    os.makedirs(REF_BUILD_DIR, exist_ok=True)
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    built = []
    for name in _CYTHON_MODULES:
        src = os.path.join(REFERENCE_ROOT, 'pb_bss', 'extraction', 'cythonized', name + '.pyx')
        c_file = os.path.join(REF_BUILD_DIR, name + '.c')
        so = os.path.join(REF_BUILD_DIR, name + ext)
        if force or not os.path.exists(so):
            subprocess.check_call([sys.executable, '-m', 'cython', '-3', src, '-o', c_file])
            subprocess.check_call([
                'gcc', '-O2', '-shared', '-fPIC', '-w',
                '-I' + sysconfig.get_paths()['include'], '-I' + numpy.get_include(),
                c_file, '-o', so])
        built.append(so)
    return built


def load_cython():
    """Inject the compiled modules as pb_bss.extraction.cythonized.{get_gev_vector, c_eig} so
    that importing pb_bss.extraction.beamformer finds c_gev_available == c_eig_available ==
    True (the import guard at extraction/beamformer.py:38-56).  Call before importing it."""
    import importlib.util
    load()
    pkg_name = 'pb_bss.extraction.cythonized'
    if pkg_name not in sys.modules:
        pkg = types.ModuleType(pkg_name)
        pkg.__path__ = [REF_BUILD_DIR]
        sys.modules[pkg_name] = pkg
    for so in build_cython():
        name = os.path.basename(so).split('.')[0]
        full = pkg_name + '.' + name
        if full in sys.modules:
            continue
        spec = importlib.util.spec_from_file_location(full, so)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        sys.modules[full] = mod
    return [sys.modules[pkg_name + '.' + n] for n in _CYTHON_MODULES]
