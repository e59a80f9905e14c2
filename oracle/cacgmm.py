"""NumPy restatement of the reference cACGMM EM path (TEST INFRASTRUCTURE).

Oracle for pb_bss.distribution.CACGMMTrainer.fit / CACGMM.predict.  Never
imported by the product package.  Citations are ``file:line`` under
/root/reference/pb_bss/.  Pinned against the real reference by
oracle/make_golden.py -> tests/golden/ (see oracle/__init__.py).

Shapes follow the reference: observations enter as (..., N, D), the working
layout is (..., D, N) ("time last"), affiliations are (..., K, N).
"""
import numpy as np

import contextlib

# Timing mode (bench.py's cpu_baseline, tools/record_reference_timings.py): issue the SAME NumPy
# calls as the reference where this restatement would otherwise be cheaper -- the E-step's
# five-operand einsum(optimize='optimal') (complex_angular_central_gaussian.py:187-196: a path
# search per call and its own temporaries) instead of the three small contractions below, the vMF
# mixture's re-normalisation of the observation on every predict (vmfmm.py:28-31,
# von_mises_fisher.py:71-73).  Same results up to rounding; a CPU baseline timed in this mode costs
# what the reference costs (checked against the reference itself in the build container:
# profiles/reference_cpu_timings.json, column oracle_reference_shaped).
REFERENCE_SHAPED = False


@contextlib.contextmanager
def reference_shaped(on=True):
    global REFERENCE_SHAPED
    before = REFERENCE_SHAPED
    REFERENCE_SHAPED = bool(on)
    try:
        yield
    finally:
        REFERENCE_SHAPED = before


__all__ = [
    'reference_shaped', 'unit_norm_where', 'normalize_observation', 'force_hermitian',
    'cacg_log_pdf', 'log_pdf_to_affiliation', 'estimate_mixture_weight',
    'cacg_from_covariance', 'cacg_m_step', 'cacg_covariance',
    'em_fit', 'em_predict', 'e_step',
]


def unit_norm_where(x, axis=-1, eps=None, ord=None):
    """distribution/utils.py:223-256 with eps_style='where': divide by the
    norm, a zero norm is replaced by ``eps`` (so all-zero vectors stay 0)."""
    nrm = np.linalg.norm(x, ord=ord, axis=axis, keepdims=True)
    nrm = np.where(nrm == 0, eps, nrm)
    return x / nrm


def normalize_observation(y):
    """distribution/complex_angular_central_gaussian.py:34-55.
    (..., N, D) -> unit-norm over D -> contiguous (..., D, N)."""
    y = unit_norm_where(y, axis=-1, eps=np.finfo(y.dtype).tiny)
    return np.ascontiguousarray(np.swapaxes(y, -2, -1))


def force_hermitian(m):
    """distribution/utils.py:318-329."""
    return (m + np.swapaxes(m.conj(), -1, -2)) / 2


def cacg_log_pdf(y, eigvec, eigval):
    """complex_angular_central_gaussian.py:167-203.

    y (..., D, N) normalised; eigvec (..., D, D); eigval (..., D).
    Returns (log_pdf, quadratic_form), both (..., N) after broadcasting.
    The reference's einsum (optimize='optimal') first forms
    B^-1 = V diag(1/lambda) V^H, then y^H (B^-1 y); we follow that order.
    """
    D = y.shape[-2]
    if REFERENCE_SHAPED:  # the reference's own call, operand for operand (:187-196)
        q = np.abs(np.einsum('...dt,...de,...e,...ge,...gt->...t', y.conj(), eigvec, 1 / eigval,
                             eigvec.conj(), y, optimize='optimal'))
    else:
        binv = np.einsum('...de,...e,...ge->...dg', eigvec, 1 / eigval, eigvec.conj())
        by = np.einsum('...dg,...gt->...dt', binv, y)
        q = np.abs(np.einsum('...dt,...dt->...t', y.conj(), by))
    q = np.maximum(q, np.finfo(y.dtype).tiny)
    log_pdf = -D * np.log(q) - np.sum(np.log(eigval), axis=-1)[..., None]
    return log_pdf, q


def log_pdf_to_affiliation(weight, log_pdf, source_activity_mask=None,
                           affiliation_eps=0.):
    """distribution/mixture_model_utils.py:7-55.  Softmax over the class axis
    (-2) with the mixture weight applied in the linear domain, optional
    activity mask, floor on the denominator and a final clip WITHOUT
    re-normalisation."""
    a = np.exp(log_pdf - np.amax(log_pdf, axis=-2, keepdims=True))
    a = a * weight
    if source_activity_mask is not None:
        a = a * source_activity_mask
    a = a / np.maximum(a.sum(axis=-2, keepdims=True), np.finfo(a.dtype).tiny)
    if affiliation_eps != 0:
        a = np.clip(a, affiliation_eps, 1 - affiliation_eps)
    return a


def estimate_mixture_weight(affiliation, saliency=None, weight_constant_axis=-1):
    """distribution/mixture_model_utils.py:133-203."""
    affiliation = np.asarray(affiliation)
    if isinstance(weight_constant_axis, int) and \
            weight_constant_axis % affiliation.ndim - affiliation.ndim == -2:
        K = affiliation.shape[-2]
        return np.full([K, 1], 1 / K)
    if isinstance(weight_constant_axis, list):
        weight_constant_axis = tuple(weight_constant_axis)
    if saliency is None:
        return np.mean(affiliation, axis=weight_constant_axis, keepdims=True)
    s = np.sum(affiliation * saliency[..., None, :],
               axis=weight_constant_axis, keepdims=True)
    return unit_norm_where(s, ord=1, axis=-2, eps=1e-10)


def cacg_from_covariance(cov, eigenvalue_floor=0., covariance_norm='eigenvalue'):
    """complex_angular_central_gaussian.py:82-132 (eigh branch)."""
    if covariance_norm == 'trace':
        tr = np.einsum('...dd', cov)[..., None, None]
        cov = cov / np.maximum(tr, np.finfo(tr.dtype).tiny)
    else:
        assert covariance_norm in ('eigenvalue', False), covariance_norm
    lam, vec = np.linalg.eigh(cov)
    lam = lam.real
    top = np.amax(lam, axis=-1, keepdims=True)
    if covariance_norm == 'eigenvalue':
        lam = lam / np.maximum(top, np.finfo(lam.dtype).tiny)
        lam = np.maximum(lam, eigenvalue_floor)
    else:
        lam = np.maximum(lam, top * eigenvalue_floor)
    assert np.isfinite(lam).all(), lam
    return vec, lam


def cacg_m_step(y, saliency, quadratic_form, hermitize=True,
                covariance_norm='eigenvalue', eigenvalue_floor=1e-10):
    """complex_angular_central_gaussian.py:253-342.

    y (..., D, N); saliency (..., N) or None; quadratic_form (..., N).
    Returns (eigvec, eigval)."""
    D = y.shape[-2]
    N = quadratic_form.shape[-1]
    if saliency is None:
        saliency = 1
        denom = np.array(N, dtype=np.float64)
    else:
        denom = np.sum(saliency, axis=-1)[..., None, None]
    q = np.maximum(quadratic_form, 10 * np.finfo(quadratic_form.dtype).tiny)
    cov = D * np.einsum('...dn,...Dn,...n->...dD', y, y.conj(), saliency / q)
    assert np.isfinite(q).all()
    cov = cov / np.maximum(denom, np.finfo(denom.dtype).tiny)
    assert np.isfinite(cov).all()
    if hermitize:
        cov = force_hermitian(cov)
    return cacg_from_covariance(cov, eigenvalue_floor=eigenvalue_floor,
                                covariance_norm=covariance_norm)


def cacg_covariance(eigvec, eigval):
    """complex_angular_central_gaussian.py:141-148:  V diag(lambda) V^H."""
    return np.einsum('...wx,...x,...zx->...wz', eigvec, eigval, eigvec.conj())


def e_step(yn, weight, eigvec, eigval, source_activity_mask=None,
           affiliation_eps=0.):
    """distribution/cacgmm.py:73-95 (CACGMM._predict) on normalised
    (..., D, N) input.  Returns (affiliation, quadratic_form, log_pdf)."""
    log_pdf, q = cacg_log_pdf(yn[..., None, :, :], eigvec, eigval)
    aff = log_pdf_to_affiliation(weight, log_pdf,
                                 source_activity_mask=source_activity_mask,
                                 affiliation_eps=affiliation_eps)
    return aff, q, log_pdf


def m_step(yn, q, aff, saliency=None, hermitize=True,
           covariance_norm='eigenvalue', eigenvalue_floor=1e-10,
           weight_constant_axis=(-1,)):
    """distribution/cacgmm.py:315-345 (CACGMMTrainer._m_step)."""
    weight = estimate_mixture_weight(aff, saliency, weight_constant_axis)
    masked = aff if saliency is None else aff * saliency[..., None, :]
    vec, lam = cacg_m_step(yn[..., None, :, :], masked, q, hermitize=hermitize,
                           covariance_norm=covariance_norm,
                           eigenvalue_floor=eigenvalue_floor)
    return weight, vec, lam


def em_fit(y, initialization, iterations=100, *, saliency=None,
           source_activity_mask=None, weight_constant_axis=(-1,),
           hermitize=True, covariance_norm='eigenvalue',
           affiliation_eps=1e-10, eigenvalue_floor=1e-10,
           return_trace=False):
    """distribution/cacgmm.py:142-280 (CACGMMTrainer.fit) for an ndarray
    affiliation initialisation (..., K, N) or a model dict
    {'weight', 'eigvec', 'eigval'}.  y is (..., N, D) complex.

    Returns dict(weight, eigvec, eigval); with return_trace also the
    per-iteration (affiliation, quadratic_form) lists.
    """
    assert np.iscomplexobj(y) and y.shape[-1] > 1 and iterations > 0
    yn = normalize_observation(y)
    *indep, D, N = yn.shape
    model = None
    if isinstance(initialization, dict):
        model = initialization
    else:
        K = initialization.shape[-2]
        shape = (*indep, K, N)
        assert initialization.ndim == len(shape)
        aff = np.broadcast_to(initialization.astype(yn.real.dtype), shape)
        q = np.ones(shape, dtype=yn.real.dtype)
    trace = []
    for _ in range(iterations):
        if model is not None:
            aff, q, _ = e_step(yn, model['weight'], model['eigvec'],
                               model['eigval'],
                               source_activity_mask=source_activity_mask,
                               affiliation_eps=affiliation_eps)
        w, vec, lam = m_step(yn, q, aff, saliency=saliency, hermitize=hermitize,
                             covariance_norm=covariance_norm,
                             eigenvalue_floor=eigenvalue_floor,
                             weight_constant_axis=weight_constant_axis)
        model = dict(weight=w, eigvec=vec, eigval=lam)
        if return_trace:
            trace.append((aff.copy(), q.copy()))
    if return_trace:
        return model, trace
    return model


def em_predict(model, y, source_activity_mask=None, return_quadratic_form=False):
    """distribution/cacgmm.py:64-71 (CACGMM.predict): re-normalise, one E-step
    with affiliation_eps = 0."""
    yn = normalize_observation(y)
    aff, q, _ = e_step(yn, model['weight'], model['eigvec'], model['eigval'],
                       source_activity_mask=source_activity_mask)
    return (aff, q) if return_quadratic_form else aff


def log_likelihood(model, y):
    """distribution/cacgmm.py:97-138: sum_t,f logsumexp_k(log_pdf) -- note the
    reference does NOT include the mixture weights here."""
    from scipy.special import logsumexp
    yn = normalize_observation(y)
    log_pdf, _ = cacg_log_pdf(yn[..., None, :, :], model['eigvec'], model['eigval'])
    return np.sum(logsumexp(log_pdf, axis=-2))
