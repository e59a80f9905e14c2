"""Drop-in for the hot-path part of pb_bss.extraction
(reference: pb_bss/extraction/__init__.py)."""
from .beamformer import (
    apply_beamforming_vector,
    apply_online_beamforming_vector,
    condition_covariance,
    distortionless_normalization,
    get_lcmv_vector,
    get_lcmv_vector_souden,
    get_mvdr_vector_merl,
    get_pca,
    mvdr_snr_postfilter,
    phase_correction,
    zero_degree_normalization,
    blind_analytic_normalization,
    get_gev_vector,
    get_mvdr_vector,
    get_mvdr_vector_souden,
    get_optimal_reference_channel,
    get_pca_vector,
    get_power_spectral_density_matrix,
    get_wmwf_vector,
    stable_solve,
)
from .beamformer_wrapper import get_bf_vector
from .beamformer_wrapper import get_bf_vector as get_single_source_bf_vector  # extraction/__init__.py:4

__all__ = [
    'get_power_spectral_density_matrix', 'get_mvdr_vector_souden',
    'get_mvdr_vector', 'get_pca_vector', 'get_gev_vector',
    'blind_analytic_normalization', 'apply_beamforming_vector',
    'get_optimal_reference_channel', 'stable_solve', 'get_bf_vector',
    'get_wmwf_vector', 'get_pca', 'get_mvdr_vector_merl', 'get_lcmv_vector',
    'get_lcmv_vector_souden', 'distortionless_normalization', 'mvdr_snr_postfilter',
    'zero_degree_normalization', 'phase_correction', 'condition_covariance',
    'apply_online_beamforming_vector', 'get_single_source_bf_vector',
]
