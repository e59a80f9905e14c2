"""pb_bss_amd -- MI355X-native engine behind pb_bss's cACGMM / beamformer API.

    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.extraction import get_power_spectral_density_matrix, get_bf_vector

The Python signatures mirror fgnt/pb_bss; the arithmetic runs in hand-written
HIP kernels (pb_bss_amd/csrc) behind the C ABI of include/pbbss.h.
"""
from . import distribution, extraction  # noqa: F401
from .distribution.utils import (arithmetic, result_dtype, set_arithmetic,  # noqa: F401
                                 set_result_dtype)

__version__ = '0.1.0'
