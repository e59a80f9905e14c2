"""NumPy/SciPy restatement of the reference complex-Watson mixture model
(TEST INFRASTRUCTURE; SURVEY.md section 8f row N2, BASELINE config 4).
Citations: /root/reference/pb_bss/distribution/{complex_watson,cwmm}.py.
Pinned by oracle/make_golden.py -> tests/golden/cwmm_*.npz.
"""
import math

import numpy as np
from scipy.interpolate import interp1d
from scipy.special import hyp1f1

from .cacgmm import estimate_mixture_weight, log_pdf_to_affiliation

__all__ = ['normalize_observation', 'watson_log_norm', 'watson_log_pdf',
           'make_spline', 'hypergeometric_ratio', 'watson_m_step', 'cwmm_fit',
           'cwmm_predict']


def normalize_observation(y):
    """complex_watson.py:16-29: (..., N, D) / max(norm, tiny); layout unchanged."""
    return y / np.maximum(np.linalg.norm(y, axis=-1, keepdims=True),
                          np.finfo(y.dtype).tiny)


def watson_log_norm(concentration, dimension):
    """complex_watson.py:157-168 (log_norm_1f1)."""
    norm = hyp1f1(1, dimension, concentration) * (
        2 * np.pi ** dimension / math.factorial(dimension - 1))
    return np.log(norm)


def watson_log_pdf(y, mode, concentration):
    """complex_watson.py:73-87: y (..., N, D) unit norm, mode (..., D),
    concentration (...) -> (..., N)."""
    r = np.einsum('...d,...d', y, mode[..., None, :].conj())
    r = r.real ** 2 + r.imag ** 2
    return r * concentration[..., None] - watson_log_norm(
        concentration, mode.shape[-1])[..., None]


def hypergeometric_ratio(concentration, dimension):
    """complex_watson.py:258-262."""
    return hyp1f1(2, dimension + 1, concentration) / (
        dimension * hyp1f1(1, dimension, concentration))


def make_spline(dimension, max_concentration=500, spline_markers=1000):
    """complex_watson.py:238-256: quadratic interp1d of the inverse ratio."""
    x = np.logspace(-3, np.log10(max_concentration), spline_markers)
    y = hypergeometric_ratio(x, dimension)
    return interp1d(y, x, kind='quadratic', assume_sorted=True, bounds_error=False,
                    fill_value=(0, max_concentration))


def watson_m_step(y, saliency, spline):
    """complex_watson.py:300-315 (+ utils.get_pca :111-167, eigh branch).
    y (..., N, D); saliency (..., N) -> (mode (..., D), concentration (...))."""
    cov = np.einsum('...n,...nd,...nD->...dD', saliency, y, y.conj())
    cov = cov / np.einsum('...n->...', saliency)[..., None, None]
    val, vec = np.linalg.eigh(cov)
    return vec[..., -1], spline(val[..., -1])


def cwmm_predict(model, y, normalized=False):
    """cwmm.py:25-52."""
    if not normalized:
        y = normalize_observation(y)
    lp = watson_log_pdf(y[..., None, :, :], model['mode'], model['concentration'])
    return log_pdf_to_affiliation(model['weight'], lp, source_activity_mask=None,
                                  affiliation_eps=0.)


def cwmm_fit(y, initialization, iterations=100, saliency=None,
             weight_constant_axis=(-1,), max_concentration=500, spline_markers=1000):
    """cwmm.py:76-182 / :217-240 for an ndarray initialisation (..., K, N)."""
    y = normalize_observation(y)
    D = y.shape[-1]
    spline = make_spline(D, max_concentration, spline_markers)
    if saliency is None:
        saliency = np.ones_like(initialization[..., 0, :])
    aff = initialization
    model = None
    for _ in range(iterations):
        if model is not None:
            aff = cwmm_predict(model, y, normalized=True)
        weight = estimate_mixture_weight(aff, saliency, weight_constant_axis)
        mode, conc = watson_m_step(y[..., None, :, :], aff * saliency[..., None, :], spline)
        model = dict(weight=weight, mode=mode, concentration=conc)
    return model
