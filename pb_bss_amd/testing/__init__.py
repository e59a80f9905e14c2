"""Synthetic input generators (NumPy only) shared by bench.py, the examples and the parity
tests; the counterpart of the reference's ``pb_bss/testing/dummy_data.py``.  No arithmetic of
the separation path lives here."""
from . import synth

__all__ = ['synth']
