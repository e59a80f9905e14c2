"""Parameter-container plumbing shared by the model dataclasses.

Mirrors pb_bss/distribution/utils.py:118-220 (`_ProbabilisticModel`): nested
to_dict / from_dict and an AttributeError that suggests close field names.
Host-side only; no numerics.
"""
import contextlib
import difflib
import os
from dataclasses import fields, is_dataclass

import numpy as np

from .. import _lib


class _ProbabilisticModel:
    def to_dict(self):
        out = {}
        for f in fields(self):
            v = getattr(self, f.name)
            out[f.name] = v.to_dict() if isinstance(v, _ProbabilisticModel) else v
        return out

    @classmethod
    def from_dict(cls, d):
        names = [f.name for f in fields(cls)]
        assert set(names) == set(d.keys()), (names, list(d.keys()))
        return cls(**d)

    def __getattr__(self, name):
        names = [f.name for f in fields(self)] if is_dataclass(self) else []
        close = difflib.get_close_matches(name, names) or names
        raise AttributeError(
            f'{self.__class__.__name__!r} object has no attribute {name!r}.\n'
            f'Close matches: {close}')


def get_trainer_class_from_model(parameter):
    """Model class (or instance) -> its trainer class, by name: `X` -> `XTrainer` looked up in
    pb_bss_amd.distribution (reference: distribution/utils.py:6-28)."""
    from .. import distribution
    cls = parameter if isinstance(parameter, type) else type(parameter)
    assert 'Trainer' not in cls.__name__, cls.__name__
    return getattr(distribution, cls.__name__ + 'Trainer')


def parameter_from_dict(parameter_class_or_str, d: dict):
    """Rebuild a model from `model.to_dict()`; the class may be given by name
    (reference: distribution/utils.py:83-113)."""
    if isinstance(parameter_class_or_str, str):
        from .. import distribution
        parameter_class_or_str = getattr(distribution, parameter_class_or_str)
    return parameter_class_or_str.from_dict(d)


def force_hermitian(matrix):
    """(A + A^H) / 2 over the last two axes (reference: distribution/utils.py:318-329); NumPy
    arrays and torch tensors alike."""
    if _lib.is_torch(matrix):
        return (matrix + matrix.conj().transpose(-1, -2)) / 2
    matrix = np.asarray(matrix)
    return (matrix + np.swapaxes(matrix.conj(), -1, -2)) / 2


def stack_parameters(parameters):
    """A list of equally-typed model objects -> ONE model whose every parameter array carries a
    new leading axis (pb_bss/distribution/utils.py:259-316: e.g. the per-utterance `CACGMM`s of a
    batch stacked into one batched model, which `predict` then serves in a single launch).
    Nested models are stacked recursively; NumPy parameters come back as NumPy arrays, device
    tensors as device tensors (the reference only knows the former); the reference's assertions
    -- one model type, one type per field -- are kept."""
    parameters = list(parameters)
    assert len(parameters) > 0, 'stack_parameters needs at least one model'
    kinds = {type(p) for p in parameters}
    assert len(kinds) == 1, kinds
    cls = kinds.pop()
    stacked = {}
    for f in fields(cls):
        values = [getattr(p, f.name) for p in parameters]
        value_kinds = {type(v) for v in values}
        assert len(value_kinds) == 1, (f.name, value_kinds)
        if is_dataclass(values[0]):
            stacked[f.name] = stack_parameters(values)
        elif _lib.is_torch(values[0]):
            stacked[f.name] = _lib.torch().stack(values)
        else:
            stacked[f.name] = np.stack(values)
    return cls(**stacked)


# ---- random affiliation initialisation (`num_classes=` instead of `initialization=`) ---------
# The reference draws np.random.uniform(size=(..., K, N)) from NumPy's GLOBAL generator and
# normalises over the classes (cacgmm.py:205-210, cwmm.py:126-131, ...).  'numpy' (default)
# consumes exactly that stream -- a seeded script sees the numbers the reference would see --
# and only moves the normalisation to the device (one host pass over the array less; the class
# sum is taken k = 0, 1, ... like the reference's einsum, so the result is bit-identical).
# 'device': the draw itself happens on the GPU (torch's Philox generator, seed with
# torch.manual_seed): ~30 us instead of ~2 ms of host RNG + 6 MB of PCIe for F=513, K=3,
# T=500 -- but NumPy's global stream is neither consumed nor reproduced.  Opt-in.
_RANDOM_INITS = ('numpy', 'device')
_random_init = os.environ.get('PBBSS_RANDOM_INIT', 'numpy')
assert _random_init in _RANDOM_INITS, _random_init


def set_random_init(mode):
    """Where the random affiliation initialisation of `fit(..., num_classes=K)` is drawn:
    'numpy' (the reference's global-RNG stream, default) or 'device' (on the GPU; a different
    stream).  Returns the previous setting."""
    global _random_init
    assert mode in _RANDOM_INITS, (mode, _RANDOM_INITS)
    old, _random_init = _random_init, mode
    return old


@contextlib.contextmanager
def random_init(mode):
    old = set_random_init(mode)
    try:
        yield
    finally:
        set_random_init(old)


def random_affiliation(shape, device):
    """(..., K, N) float64 device tensor, uniform draws normalised over the class axis."""
    t = _lib.torch()
    if _random_init == 'device':
        aff = t.rand(tuple(shape), dtype=t.float64, device=device)
    else:
        aff = t.from_numpy(np.random.uniform(size=tuple(shape))).to(device)
    den = aff[..., 0, :].clone()
    for k in range(1, shape[-2]):  # ascending class order, like einsum('...kn->...n')
        den += aff[..., k, :]
    return aff / den[..., None, :]


def as_result(x, like_torch):
    """Device tensor -> what the caller works with (torch in, torch out;
    NumPy in, NumPy out, as the reference returns)."""
    if x is None:
        return None
    return x if like_torch else _lib.to_host(x)


# ---- result dtypes -----------------------------------------------------------------------
# The device arithmetic is float64 throughout.  The reference computes in the precision of
# its operands (cacgmm.py:226-227: an ndarray initialisation is cast to y.real.dtype, so
# complex64 observations give a float32 / complex64 model and float32 masks).  'reference'
# rounds the RESULTS to those dtypes; 'float64' (default) returns the working precision.
_RESULT_DTYPES = ('float64', 'reference')
_result_dtype = os.environ.get('PBBSS_RESULT_DTYPE', 'float64')
assert _result_dtype in _RESULT_DTYPES, _result_dtype


def set_result_dtype(mode):
    """'float64': every result in the device's working precision; 'reference': the dtypes
    the reference returns for the same operands (single precision results for complex64
    observations with an array / single-precision-model initialisation).  Returns the
    previous setting."""
    global _result_dtype
    assert mode in _RESULT_DTYPES, (mode, _RESULT_DTYPES)
    old, _result_dtype = _result_dtype, mode
    return old


@contextlib.contextmanager
def result_dtype(mode):
    old = set_result_dtype(mode)
    try:
        yield
    finally:
        set_result_dtype(old)


# ---- arithmetic of the cACGMM trainer ------------------------------------------------------
# 'float64' (default): the float64 kernel for every input.  'reference': where the reference
# itself computes in single precision -- a complex64 observation with an array initialisation
# (cacgmm.py:226-227) -- use the packed-FP32 kernel (csrc/cacgmm_em32.hpp: E / M phases in
# float32, class sums and factorisation in float64); anything that kernel does not serve
# (complex128, long utterances, bin-coupled weights) runs in float64.
_ARITHMETICS = ('float64', 'reference')
_arithmetic = os.environ.get('PBBSS_ARITHMETIC', 'float64')
assert _arithmetic in _ARITHMETICS, _arithmetic


def set_arithmetic(mode):
    """Select the arithmetic of `CACGMMTrainer.fit`; returns the previous setting."""
    global _arithmetic
    assert mode in _ARITHMETICS, (mode, _ARITHMETICS)
    old, _arithmetic = _arithmetic, mode
    return old


@contextlib.contextmanager
def arithmetic(mode):
    old = set_arithmetic(mode)
    try:
        yield
    finally:
        set_arithmetic(old)


def reference_arithmetic():
    return _arithmetic == 'reference'


def _is_single(x):
    """float32 / complex64 operand (NumPy or torch)?"""
    return str(x.dtype).rsplit('.', 1)[-1] in ('float32', 'complex64', 'float16')


def reference_single(*operands):
    """True when the reference would carry these operands (None = absent) in single
    precision and the 'reference' result dtype is selected."""
    if _result_dtype != 'reference':
        return False
    return all(_is_single(x) for x in operands if x is not None)


def to_single(x):
    """float64 -> float32 / complex128 -> complex64 (NumPy or torch; None passes)."""
    if x is None:
        return None
    if _lib.is_torch(x):
        t = _lib.torch()
        return x.to(t.complex64 if x.is_complex() else t.float32)
    x = np.asarray(x)
    return x.astype(np.complex64 if np.iscomplexobj(x) else np.float32)
