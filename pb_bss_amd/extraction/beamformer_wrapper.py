"""String-keyed beamformer dispatcher.

Mirrors pb_bss/extraction/beamformer_wrapper.py:117-236 (`get_bf_vector`) and
its rank-1 / ATF helpers (:11-104) on top of the device functions in
`beamformer.py`.  Cores on the hot path: 'pca', 'gev', 'mvdr_souden',
'pca+mvdr', 'scaled_gev_atf+mvdr', 'wmwf', the 'rank1_pca+...' / 'rank1_gev+...'
variants and 'ch<N>', each optionally followed by '+ban'.
"""
import numpy as np

from .. import _lib
from .beamformer import (
    blind_analytic_normalization,
    get_gev_vector,
    get_mvdr_vector,
    get_mvdr_vector_souden,
    get_pca_vector,
    get_wmwf_vector,
)

__all__ = ['get_bf_vector']


def _xp(x):
    return _lib.torch() if _lib.is_torch(x) else np


def _outer(a):
    xp = _xp(a)
    return xp.einsum('...d,...e->...de', a, a.conj())


def _trace(m):
    xp = _xp(m)
    return xp.einsum('...dd->...', m) if xp is not np else np.trace(m, axis1=-1, axis2=-2)


def _rank_one(covariance_matrix, a):
    """Scale a a^H to the trace of the covariance (wrapper.py:18-25, :61-69)."""
    r1 = _outer(a)
    scale = _trace(covariance_matrix) / _trace(r1)
    return scale[..., None, None] * r1


def _gev_atf_vector(covariance_matrix, noise_covariance_matrix, **gev_kwargs):
    """Phi_nn w_gev as an ATF estimate (wrapper.py:28-48)."""
    assert noise_covariance_matrix is not None
    w = get_gev_vector(covariance_matrix, noise_covariance_matrix, **gev_kwargs)
    return _xp(w).einsum('...de,...e->...d', _match(noise_covariance_matrix, w), w)


def _match(x, like):
    """Bring x to the array family / dtype of `like`."""
    if _lib.is_torch(like):
        return _lib.to_device(x, like.dtype)
    return np.asarray(x)


def _atf_vector(atf_type, target_psd_matrix, noise_psd_matrix, **atf_kwargs):
    if atf_type == 'pca':
        return get_pca_vector(target_psd_matrix, **atf_kwargs)
    if atf_type == 'scaled_gev_atf':
        return _gev_atf_vector(target_psd_matrix, noise_psd_matrix, **atf_kwargs)
    raise ValueError(atf_type, 'use either pca or scaled_gev_atf')


def _rank_1_approximation(atf_type, target_psd_matrix, noise_psd_matrix, **atf_kwargs):
    if atf_type == 'rank1_pca':
        a = get_pca_vector(target_psd_matrix, **atf_kwargs)
    elif atf_type == 'rank1_gev':
        a = _gev_atf_vector(target_psd_matrix, noise_psd_matrix, **atf_kwargs)
    else:
        raise ValueError(atf_type, 'use either rank1_pca or rank1_gev')
    return _rank_one(_match(target_psd_matrix, a), a)


def get_bf_vector(beamformer, target_psd_matrix, noise_psd_matrix=None, **bf_kwargs):
    """Light wrapper to obtain a beamforming vector, e.g. 'mvdr_souden',
    'gev+ban', 'rank1_gev+mvdr_souden+ban'.  Steps are separated by '+';
    options for the ATF / rank-1 step go in bf_kwargs['atf_kwargs']."""
    assert 'lcmv' not in beamformer, (
        'Since the LCMV beamformer and its variants sufficiently differ from '
        'all other beamforming approaches, we provide a separate wrapper '
        'function `get_multi_source_bf_vector()`.'
    )
    assert isinstance(beamformer, str), beamformer
    ban = beamformer.endswith('+ban')
    core = beamformer[:-len('+ban')] if ban else beamformer

    if core == 'pca':
        w = get_pca_vector(target_psd_matrix, **bf_kwargs)
    elif core in ['pca+mvdr', 'scaled_gev_atf+mvdr']:
        atf, _ = core.split('+')
        atf_vector = _atf_vector(atf, target_psd_matrix, noise_psd_matrix,
                                 **bf_kwargs.pop('atf_kwargs', {}))
        w = get_mvdr_vector(atf_vector, noise_psd_matrix)
    elif core in ['mvdr_souden', 'rank1_pca+mvdr_souden', 'rank1_gev+mvdr_souden']:
        if core != 'mvdr_souden':
            rank1_type, _ = core.split('+')
            target_psd_matrix = _rank_1_approximation(
                rank1_type, target_psd_matrix, noise_psd_matrix,
                **bf_kwargs.pop('atf_kwargs', {}))
        w = get_mvdr_vector_souden(target_psd_matrix, noise_psd_matrix, **bf_kwargs)
    elif core in ['gev', 'rank1_pca+gev', 'rank1_gev+gev']:
        if core != 'gev':
            rank1_type, _ = core.split('+')
            target_psd_matrix = _rank_1_approximation(
                rank1_type, target_psd_matrix, noise_psd_matrix,
                **bf_kwargs.pop('atf_kwargs', {}))
        w = get_gev_vector(target_psd_matrix, noise_psd_matrix, **bf_kwargs)
    elif core in ['wmwf', 'rank1_pca+wmwf', 'rank1_gev+wmwf']:
        if core != 'wmwf':
            rank1_type, _ = core.split('+')
            target_psd_matrix = _rank_1_approximation(
                rank1_type, target_psd_matrix, noise_psd_matrix,
                **bf_kwargs.pop('atf_kwargs', {}))
        w = get_wmwf_vector(target_psd_matrix, noise_psd_matrix, **bf_kwargs)
    elif 'ch' in core and core[2:].isdigit():
        D = target_psd_matrix.shape[-1]
        w = np.zeros(D)
        w[int(core[2:])] = 1
        w = np.broadcast_to(w, tuple(target_psd_matrix.shape[:-1]))
        if _lib.is_torch(target_psd_matrix):
            w = _lib.to_device(np.ascontiguousarray(w))
    else:
        raise ValueError(
            f'Could not find implementation for {core}.\n'
            f'Original call contained {beamformer}.')
    if ban:
        w = blind_analytic_normalization(w, noise_psd_matrix)
    return w
