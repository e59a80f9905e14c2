"""Step-wise EM loop of the real-embedding mixtures (VMFMM, GMM) on the device, for the options
the fused loops (`pbbss_vmfmm_fit`, `pbbss_gmm_fit`, `pbbss_gmm_full_fit`) do not carry:
`weight_constant_axis` sets beyond (-1,) / -2 -- weights shared over independent axes,
frame-varying weights -- and `covariance_type='diagonal'`.

It is the reference's loop statement by statement (vmfmm.py:131-172, gmm.py:121-171):

    affiliation = predict(model)                  -> class log-pdfs + log_pdf_to_affiliation
    weight      = estimate_mixture_weight(...)    -> pbbss_estimate_mixture_weight
    component   = Trainer()._fit(y, affiliation * saliency)

with every step a device kernel (`pbbss_embed_log_pdf` / `pbbss_gauss_full_log_pdf`,
`pbbss_log_pdf_to_affiliation`, `pbbss_estimate_mixture_weight`, `pbbss_embed_fit` /
`pbbss_gauss_full_fit`); nothing returns to the host inside the loop.
"""
import numpy as np

from .. import _lib, engine

KINDS = {'vmf': _lib.EMBED_VMF, 'spherical': _lib.EMBED_GAUSS_SPHERICAL,
         'diagonal': _lib.EMBED_GAUSS_DIAG, 'full': _lib.EMBED_GAUSS_FULL}


def device_weight(aff, sal, weight_constant_axis, indep):
    """estimate_mixture_weight (mixture_model_utils.py:133-203), reference-shaped (keepdims),
    as a device tensor."""
    from .cacgmm import CACGMMTrainer
    from .mixture_model_utils import _host_estimate_mixture_weight
    t = _lib.torch()
    nd = len(indep) + 2
    K = aff.shape[-2]
    if isinstance(weight_constant_axis, int) and weight_constant_axis % nd - nd == -2:
        return t.full((K, 1), 1.0 / K, dtype=t.float64, device=aff.device)  # :180-183
    w = CACGMMTrainer._device_weight(aff, sal, weight_constant_axis, indep)
    if w is None:  # axis sets the reduction kernel does not cover (e.g. the class axis in a tuple)
        w = _lib.to_device(_host_estimate_mixture_weight(
            _lib.to_host(aff), None if sal is None else _lib.to_host(sal), weight_constant_axis),
            t.float64).to(aff.device)
    return w


def log_pdf(kind, yb, mean, scale):
    """Class log-pdfs (B, K, N) of the components (mean (B,K,E), scale by kind)."""
    if kind == 'full':
        lp, st = engine.gauss_full_log_pdf(yb, mean, scale)
        if int(st.item()) != 0:
            raise ValueError(  # sklearn's _compute_precision_cholesky via gaussian.py:26
                'Fitting the mixture model failed because some components have ill-defined '
                'empirical covariance (not positive definite)')
        return lp
    if kind == 'diagonal':
        # the reference's DiagonalGaussian.log_pdf takes the (K, E) precisions of ONE mixture as a
        # K x E matrix shared by its classes (gaussian.py:87-91): one call per independent mixture
        return _lib.torch().cat([engine.embed_log_pdf(yb[b:b + 1], KINDS[kind], mean[b:b + 1],
                                                      scale[b:b + 1]) for b in range(yb.shape[0])])
    return engine.embed_log_pdf(yb, KINDS[kind], mean, scale)


def affiliation(kind, yb, mean, scale, weight, indep, K, N):
    """predict(): softmax of the class log-pdfs with a reference-shaped weight array."""
    lp = log_pdf(kind, yb, mean, scale)
    t = _lib.torch()
    w = weight.to(t.float64).to(yb.device)
    # broadcast the (possibly lower-rank, keepdims) weight against (*indep, K, N), then flatten
    # the independent axes: singleton axes stay singleton unless the array really varies there
    full = (*indep, K, N)
    while w.ndim < len(full):
        w = w.unsqueeze(0)
    lead = w.shape[:-2]
    if any(a != 1 for a in lead):
        w = w.expand(*indep, *w.shape[-2:]).reshape(-1, *w.shape[-2:])
    else:
        w = w.reshape(1, *w.shape[-2:])
    return engine.log_pdf_to_affiliation(lp, w)


def fit(kind, y, gamma0, iterations, saliency, weight_constant_axis, *, fixed_scale=None,
        min_concentration=1e-10, max_concentration=500.):
    """y (*indep, N, E) device tensor, gamma0 (*indep, K, N) float64 device tensor.
    -> dict(mean (*indep,K,E), scale, weight (reference shape)) of device tensors."""
    t = _lib.torch()
    *indep, N, E = y.shape
    indep = tuple(indep)
    K = gamma0.shape[-2]
    yb = y.reshape(-1, N, E)
    B = yb.shape[0]
    sal = None
    if saliency is not None:
        sal = _lib.to_device(saliency, t.float64).to(y.device).expand(*indep, N).contiguous()
    aff = gamma0.contiguous()
    mean = scale = weight = None
    for it in range(iterations):
        if it > 0:
            aff = affiliation(kind, yb, mean, scale, weight, indep, K, N).reshape(*indep, K, N)
        weight = device_weight(aff, sal, weight_constant_axis, indep)
        masked = aff if sal is None else aff * sal[..., None, :]
        wts = masked.reshape(B, K, N).contiguous()
        if kind == 'full':
            mean, scale = engine.gauss_full_fit(yb, wts)
        else:
            mean, scale = engine.embed_fit(yb, KINDS[kind], wts, normalize=(kind == 'vmf'),
                                           min_concentration=min_concentration,
                                           max_concentration=max_concentration)
        if fixed_scale is not None:
            scale = fixed_scale.reshape(scale.shape)
    return dict(mean=mean, scale=scale, weight=weight)
