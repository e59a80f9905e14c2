"""Synthetic inputs of the parity tests: alias of ``pb_bss_amd.testing.synth`` (the generator
lives with the package so that bench.py and the examples draw their inputs without importing
anything from ``oracle/``)."""
import sys

from pb_bss_amd.testing import synth as _synth

sys.modules[__name__] = _synth
