"""String-keyed beamformer dispatcher.

Mirrors pb_bss/extraction/beamformer_wrapper.py:117-236 (`get_bf_vector`) and
its rank-1 / ATF helpers (:11-104) on top of the device functions in
`beamformer.py`.  Cores on the hot path: 'pca', 'gev', 'mvdr_souden',
'pca+mvdr', 'scaled_gev_atf+mvdr', 'wmwf', the 'rank1_pca+...' / 'rank1_gev+...'
variants and 'ch<N>', each optionally followed by '+ban'.
"""
import numpy as np

from .. import _lib
from ..utils import labels_to_one_hot  # noqa: F401  (the reference module re-exports it)
from .beamformer import *  # noqa: F401,F403  (as the reference: beamformer_wrapper.py:4)
from .beamformer import (
    blind_analytic_normalization,
    get_gev_vector,
    get_mvdr_vector,
    get_mvdr_vector_souden,
    get_pca_vector,
    get_wmwf_vector,
)

__all__ = ['get_bf_vector']


def _on_device(fn, like, *arrays):
    """Run the device function `fn` on (N, ...)-flattened complex128 device copies of `arrays`
    and hand the result back in the array family of `like` (NumPy in -> NumPy out)."""
    t = _lib.torch()
    dev = [_lib.to_device(a, t.complex128) for a in arrays]
    out = fn(*dev)
    return out if _lib.is_torch(like) else _lib.to_host(out)


def _rank_one(covariance_matrix, a):
    """Scale a a^H to the trace of the covariance (wrapper.py:18-25, :61-69) --
    pbbss_rank_one_approximation, one thread per matrix entry."""
    from .. import engine
    lead, D = tuple(a.shape[:-1]), a.shape[-1]

    def run(cov, vec):
        cov = cov.expand(*lead, D, D).reshape(-1, D, D).contiguous()
        return engine.rank_one_approximation(cov, vec.reshape(-1, D).contiguous()).reshape(*lead, D, D)
    return _on_device(run, a, covariance_matrix, a)


def _gev_atf_vector(covariance_matrix, noise_covariance_matrix, **gev_kwargs):
    """Phi_nn w_gev as an ATF estimate (wrapper.py:28-48) -- pbbss_matvec."""
    from .. import engine
    assert noise_covariance_matrix is not None
    w = get_gev_vector(covariance_matrix, noise_covariance_matrix, **gev_kwargs)
    lead, D = tuple(w.shape[:-1]), w.shape[-1]

    def run(noise, vec):
        noise = noise.expand(*lead, D, D).reshape(-1, D, D).contiguous()
        return engine.matvec(noise, vec.reshape(-1, D).contiguous()).reshape(*lead, D)
    return _on_device(run, w, noise_covariance_matrix, w)


def get_pca_rank_one_estimate(covariance_matrix, **atf_kwargs):
    """The covariance as the outer product of its dominant eigenvector, scaled to the
    covariance's trace (wrapper.py:11-25; Wang et al., "Rank-1 constrained ...", eq. 25-26)."""
    a = get_pca_vector(covariance_matrix, **atf_kwargs)
    return _rank_one(_match(covariance_matrix, a), a)


def get_gev_rank_one_estimate(covariance_matrix, noise_covariance_matrix, **gev_kwargs):
    """The covariance as the outer product of the GEV ATF estimate Phi_nn w_gev, scaled to the
    covariance's trace (wrapper.py:49-69)."""
    a = _gev_atf_vector(covariance_matrix, noise_covariance_matrix, **gev_kwargs)
    return _rank_one(_match(covariance_matrix, a), a)


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
        return get_pca_rank_one_estimate(target_psd_matrix, **atf_kwargs)
    if atf_type == 'rank1_gev':
        return get_gev_rank_one_estimate(target_psd_matrix, noise_psd_matrix, **atf_kwargs)
    raise ValueError(atf_type, 'use either rank1_pca or rank1_gev')


def _get_response_vector(source_index, num_sources, epsilon=0.):
    """One-hot LCMV response vector, floored at `epsilon` (wrapper.py:106-114)."""
    response = labels_to_one_hot(np.array(source_index), num_sources, dtype=np.float64)
    return np.clip(response, epsilon, 1.)


# the reference's private spellings (wrapper.py:28, :72, :93, :106), for callers that import them
_get_gev_atf_vector = _gev_atf_vector
_get_atf_vector = _atf_vector
_get_rank_1_approximation = _rank_1_approximation


# filter stage -> (device function, needs the noise PSD, may be preceded by a rank-1 stage)
_FILTER_STAGE = {
    'mvdr_souden': (get_mvdr_vector_souden, True),
    'gev': (get_gev_vector, True),
    'wmwf': (get_wmwf_vector, True),
}
_ATF_STAGE = ('pca', 'scaled_gev_atf')       # '<atf>+mvdr'
_RANK1_STAGE = ('rank1_pca', 'rank1_gev')    # '<rank1>+<filter>'


def _parse(core):
    """'pca' | 'ch<N>' | '<atf>+mvdr' | ['<rank1>+']<filter>  ->  (kind, pre-stage, filter)."""
    parts = core.split('+')
    if parts == ['pca']:
        return 'pca', None, None
    if len(parts) == 1 and 'ch' in core and core[2:].isdigit():
        return 'channel', int(core[2:]), None
    if len(parts) == 2 and parts[1] == 'mvdr' and parts[0] in _ATF_STAGE:
        return 'atf_mvdr', parts[0], None
    if parts[-1] in _FILTER_STAGE and (
            len(parts) == 1 or (len(parts) == 2 and parts[0] in _RANK1_STAGE)):
        return 'filter', parts[0] if len(parts) == 2 else None, parts[-1]
    return None, None, None


def get_bf_vector(beamformer, target_psd_matrix, noise_psd_matrix=None, **bf_kwargs):
    """Light wrapper to obtain a beamforming vector from a '+'-separated recipe, e.g.
    'mvdr_souden', 'gev+ban', 'rank1_gev+mvdr_souden+ban', 'scaled_gev_atf+mvdr', 'ch0'
    (reference: beamformer_wrapper.py:117-236).  Options of the ATF / rank-1 stage go in
    bf_kwargs['atf_kwargs'], everything else to the filter stage."""
    assert 'lcmv' not in beamformer, (
        'Since the LCMV beamformer and its variants sufficiently differ from '
        'all other beamforming approaches, we provide a separate wrapper '
        'function `get_multi_source_bf_vector()`.'
    )
    assert isinstance(beamformer, str), beamformer
    ban = beamformer.endswith('+ban')
    core = beamformer[:-len('+ban')] if ban else beamformer
    kind, pre, filt = _parse(core)
    if kind is None:
        raise ValueError(
            f'Could not find implementation for {core}.\n'
            f'Original call contained {beamformer}.')
    if kind == 'pca':
        w = get_pca_vector(target_psd_matrix, **bf_kwargs)
    elif kind == 'channel':
        D = target_psd_matrix.shape[-1]
        w = np.zeros(D)
        w[pre] = 1
        w = np.broadcast_to(w, tuple(target_psd_matrix.shape[:-1]))
        if _lib.is_torch(target_psd_matrix):
            w = _lib.to_device(np.ascontiguousarray(w))
    elif kind == 'atf_mvdr':
        atf = _atf_vector(pre, target_psd_matrix, noise_psd_matrix,
                          **bf_kwargs.pop('atf_kwargs', {}))
        w = get_mvdr_vector(atf, noise_psd_matrix)
    else:
        if pre is not None:
            target_psd_matrix = _rank_1_approximation(
                pre, target_psd_matrix, noise_psd_matrix, **bf_kwargs.pop('atf_kwargs', {}))
        w = _FILTER_STAGE[filt][0](target_psd_matrix, noise_psd_matrix, **bf_kwargs)
    if ban:
        w = blind_analytic_normalization(w, noise_psd_matrix)
    return w
