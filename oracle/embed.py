"""NumPy restatement of the reference's real-embedding models (TEST INFRASTRUCTURE).

Oracle for SURVEY.md section 8(f) rows N2 (von-Mises-Fisher mixture) and N3
(joint spatial + spectral mixtures): pb_bss.distribution.von_mises_fisher,
vmfmm, gaussian, gcacgmm, vmfcacgmm.  Never imported by the product package.
Citations are ``file:line`` under /root/reference/pb_bss/distribution/.  Pinned
against the real reference by oracle/make_golden.py -> tests/golden/embed_*.npz.
"""
import itertools

import numpy as np
from scipy.special import ive

from . import cacgmm as oc

_TINY = np.finfo(np.float64).tiny


def unit_rows(y):
    """von_mises_fisher.py:71-73, vmfmm.py:76-78, gcacgmm.py:178-181."""
    y = np.asarray(y)
    return y / np.maximum(np.linalg.norm(y, axis=-1, keepdims=True), np.finfo(y.dtype).tiny)


# ---------------------------------------------------------------- von Mises-Fisher
def vmf_log_norm(concentration, dimension):
    """von_mises_fisher.py:33-44."""
    nu = dimension / 2 - 1
    return (dimension / 2 * np.log(2 * np.pi) + np.log(ive(nu, concentration))
            + (np.abs(concentration) - nu * np.log(concentration)))


def vmf_log_pdf(y, mean, concentration, normalized=False):
    """von_mises_fisher.py:62-78.  y (..., N, D), mean (..., D) -> (..., N)."""
    if not normalized:
        y = unit_rows(y)
    if oc.REFERENCE_SHAPED:  # the reference's contraction (von_mises_fisher.py:74)
        proj = np.einsum('...d,...d', y, mean[..., None, :])
    else:
        proj = np.einsum('...nd,...d->...n', y, mean)
    return proj * concentration[..., None] - vmf_log_norm(concentration, mean.shape[-1])[..., None]


def vmf_fit(y, saliency, min_concentration=1e-10, max_concentration=500):
    """von_mises_fisher.py:119-144 (Banerjee 2005 eq. 2.4, 2.5, 4.4); y unit rows."""
    dim = y.shape[-1]
    if saliency is None:
        saliency = np.ones(y.shape[:-1])
    resultant = np.einsum('...n,...nd->...d', saliency, y)
    length = np.linalg.norm(resultant, axis=-1)
    mean = resultant / np.maximum(length, np.finfo(y.dtype).tiny)[..., None]
    rbar = length / np.sum(saliency, axis=-1)
    conc = (rbar * dim - rbar ** 3) / (1 - rbar ** 2)
    return mean, np.clip(conc, min_concentration, max_concentration)


def vmfmm_predict(model, y, normalized=False):
    """vmfmm.py:19-37."""
    if not normalized:
        y = unit_rows(y)
    lp = vmf_log_pdf(y[..., None, :, :], model['mean'], model['concentration'],
                     normalized=not oc.REFERENCE_SHAPED)
    return oc.log_pdf_to_affiliation(model['weight'], lp)


def vmfmm_m_step(y, affiliation, saliency, weight_constant_axis=(-1,),
                 min_concentration=1e-10, max_concentration=500):
    """vmfmm.py:150-172."""
    weight = oc.estimate_mixture_weight(affiliation, saliency, weight_constant_axis)
    mean, conc = vmf_fit(y[..., None, :, :], affiliation * saliency[..., None, :],
                         min_concentration, max_concentration)
    return dict(weight=weight, mean=mean, concentration=conc)


def vmfmm_fit(y, initialization, iterations=100, saliency=None, weight_constant_axis=(-1,),
              min_concentration=1e-10, max_concentration=500):
    """vmfmm.py:42-148 with a given initialization."""
    y = unit_rows(y)
    if saliency is None:
        saliency = np.ones_like(initialization[..., 0, :])
    aff = initialization
    model = None
    for _ in range(iterations):
        if model is not None:
            # timing mode: VMFMM.predict normalises the (already normalised) rows again, and so
            # does VonMisesFisher.log_pdf inside it (vmfmm.py:28-31, von_mises_fisher.py:71-73)
            aff = vmfmm_predict(model, y, normalized=not oc.REFERENCE_SHAPED)
        model = vmfmm_m_step(y, aff, saliency, weight_constant_axis,
                             min_concentration, max_concentration)
    return model


# ---------------------------------------------------------------- Gaussians
def gaussian_fit(y, saliency, covariance_type='spherical'):
    """gaussian.py:152-193.  y (..., N, D), saliency (..., N) or None."""
    dim = y.shape[-1]
    if saliency is None:
        den = np.array(y.shape[-2], dtype=np.float64)
        mean = np.einsum('...nd->...d', y)
    else:
        den = np.maximum(np.einsum('...n->...', saliency), np.finfo(y.dtype).tiny)
        mean = np.einsum('...n,...nd->...d', saliency, y)
    mean = mean / den[..., None]
    diff = y - mean[..., None, :]
    w = np.ones(y.shape[:-1]) if saliency is None else saliency
    if covariance_type == 'full':
        cov = np.einsum('...n,...nd,...nD->...dD', w, diff, diff) / den[..., None, None]
    elif covariance_type == 'diagonal':
        cov = np.einsum('...n,...nd,...nd->...d', w, diff, diff) / den[..., None]
    elif covariance_type == 'spherical':
        cov = np.einsum('...n,...nd,...nd->...', w, diff, diff) / (den * dim)
    else:
        raise ValueError(f"Unknown covariance type '{covariance_type}'.")
    return mean, cov


def gaussian_log_pdf(y, mean, covariance, covariance_type='spherical'):
    """gaussian.py:35-56, 76-97, 116-137 with sklearn's precision Cholesky
    (sklearn/mixture/_gaussian_mixture.py: 1/sqrt(cov) for 'diag'; inverse
    transposed Cholesky factor for 'full') and log-determinant."""
    dim = mean.shape[-1]
    diff = y - mean[..., None, :]
    if covariance_type == 'spherical':
        pc = 1.0 / np.sqrt(covariance)
        white = pc[..., None, None] * diff
        logdet = dim * np.log(pc)
    elif covariance_type == 'diagonal':
        # NB reference quirk (gaussian.py:87-91): the (K, D) precision "Cholesky" is fed to
        # einsum as '...dD', i.e. as ONE K x D matrix shared by all classes, so
        # white[k, n, j] = sum_D pc[j, D] diff[k, n, D] -- restated as written, not as intended.
        pc = 1.0 / np.sqrt(covariance)
        white = np.einsum('...dD,...nD->...nd', pc, diff)
        logdet = np.sum(np.log(pc), axis=-1)
    elif covariance_type == 'full':
        chol = np.linalg.cholesky(covariance)
        pc = np.swapaxes(np.linalg.inv(chol), -1, -2)      # P = L^-T, cov^-1 = P P^T
        white = np.einsum('...dD,...nD->...nd', pc, diff)  # gaussian.py:46-50 (as written there)
        logdet = np.sum(np.log(np.diagonal(pc, axis1=-2, axis2=-1)), axis=-1)
    else:
        raise ValueError(covariance_type)
    return (-0.5 * dim * np.log(2 * np.pi) + logdet[..., None]
            - 0.5 * np.einsum('...nd,...nd->...n', white, white))


# ---------------------------------------------------------------- Gaussian mixture
def gmm_predict(model, y, covariance_type='spherical'):
    """gmm.py:21-25."""
    lp = gaussian_log_pdf(y[..., None, :, :], model['mean'], model['covariance'], covariance_type)
    return oc.log_pdf_to_affiliation(model['weight'], lp)


def gmm_fit(y, initialization, iterations=100, saliency=None, weight_constant_axis=(-1,),
            fixed_covariance=None, covariance_type='spherical'):
    """GMMTrainer.fit, gmm.py:33-171, given initialization ('spherical' or 'full')."""
    if saliency is None:
        saliency = np.ones_like(initialization[..., 0, :])  # :79-80
    aff = initialization
    model = None
    for _ in range(iterations):
        if model is not None:
            aff = gmm_predict(model, y, covariance_type)  # :129-130
        weight = oc.estimate_mixture_weight(aff, saliency, weight_constant_axis)  # :152-156
        mean, cov = gaussian_fit(y[..., None, :, :], aff * saliency[..., None, :],
                                 covariance_type)  # :158-162
        if fixed_covariance is not None:
            assert fixed_covariance.shape == cov.shape
            cov = fixed_covariance  # :164-171
        model = dict(weight=weight, mean=mean, covariance=cov)
    return model


# ---------------------------------------------------------------- joint models
def _unsqueeze(array, axis):
    """pb_bss/utils.py:306-335."""
    array = np.array(array)
    shape = list(array.shape)
    nd = len(shape) + len(axis)
    for p in sorted(a % nd for a in axis):
        shape.insert(p, 1)
    return array.reshape(shape)


def inline_pa_affiliation(weight, spatial_log_pdf, spectral_log_pdf, affiliation_eps=0.,
                          source_activity_mask=None):
    """mixture_model_utils.py:58-130 (per-frequency best class permutation of the
    spatial model against the spectral model; the activity mask only enters the final
    posterior, :121-126)."""
    F, K, T = spatial_log_pdf.shape
    out = np.zeros((F, K, T))
    wfull = np.broadcast_to(weight, spatial_log_pdf.shape)
    for f in range(F):
        best, best_val = None, -np.inf
        for perm in itertools.permutations(range(K)):
            lp = spatial_log_pdf[f, list(perm), :] + spectral_log_pdf[f]
            cand = np.exp(lp - lp.max(axis=-2, keepdims=True))
            cand /= np.maximum(cand.sum(axis=-2, keepdims=True), _TINY)
            val = np.sum(cand * lp)
            if val > best_val:
                best, best_val = list(perm), val
        out[f] = oc.log_pdf_to_affiliation(
            wfull[f], spatial_log_pdf[f, best, :] + spectral_log_pdf[f],
            source_activity_mask=None if source_activity_mask is None else source_activity_mask[f],
            affiliation_eps=affiliation_eps)
    return out


def joint_weight(masked_affiliation, weight_constant_axis):
    """gcacgmm.py:288-295 == vmfcacgmm.py:258-265."""
    K = masked_affiliation.shape[-2]
    if -2 in weight_constant_axis:
        return 1 / K
    w = np.sum(masked_affiliation, axis=tuple(weight_constant_axis), keepdims=True)
    w = w / np.sum(w, axis=-2, keepdims=True)
    return np.squeeze(w, axis=tuple(weight_constant_axis))


def joint_predict(model, yn, embedding, affiliation_eps=0., inline_permutation_alignment=False):
    """gcacgmm.py:66-117 / vmfcacgmm.py:57-97.  yn (F,T,D) unit-norm observation,
    embedding (F,T,E) (unit rows for the vMF variant).  -> affiliation, quadratic form (F,K,T)."""
    F, T, _ = yn.shape
    E = embedding.shape[-1]
    cacg_lp, q = oc.cacg_log_pdf(np.swapaxes(yn[..., None, :, :], -1, -2),
                                 model['eigvec'], model['eigval'])
    flat = embedding.reshape(1, F * T, E)
    if model['kind'] == 'gaussian':
        slp = gaussian_log_pdf(flat, model['mean'], model['covariance'], model['covariance_type'])
    else:
        slp = vmf_log_pdf(flat, model['mean'], model['concentration'])  # normalises again (idempotent)
    K = slp.shape[0]
    slp = slp.reshape(K, F, T).transpose(1, 0, 2)
    weight = _unsqueeze(model['weight'], model['weight_constant_axis'])
    if inline_permutation_alignment:
        aff = inline_pa_affiliation(weight, model['spatial_weight'] * cacg_lp,
                                    model['spectral_weight'] * slp, affiliation_eps)
    else:
        aff = oc.log_pdf_to_affiliation(
            weight, model['spatial_weight'] * cacg_lp + model['spectral_weight'] * slp,
            affiliation_eps=affiliation_eps)
    return aff, q


def joint_m_step(kind, yn, embedding, q, aff, saliency, *, hermitize=True,
                 covariance_norm='eigenvalue', eigenvalue_floor=1e-10,
                 covariance_type='spherical', fixed_covariance=None,
                 min_concentration=1e-10, max_concentration=500,
                 weight_constant_axis=(-1,), spatial_weight=1., spectral_weight=1.):
    """gcacgmm.py:267-333 / vmfcacgmm.py:237-301."""
    F, T, _ = yn.shape
    E = embedding.shape[-1]
    K = aff.shape[1]
    masked = aff * saliency[..., None, :]
    weight = joint_weight(masked, weight_constant_axis)
    flat = embedding.reshape(1, F * T, E)
    masked_flat = masked.transpose(1, 0, 2).reshape(K, F * T)
    model = dict(kind=kind, weight=weight, weight_constant_axis=tuple(weight_constant_axis),
                 spatial_weight=spatial_weight, spectral_weight=spectral_weight)
    if kind == 'gaussian':
        mean, cov = gaussian_fit(flat, masked_flat, covariance_type)
        if fixed_covariance is not None:
            cov = fixed_covariance
        model.update(mean=mean, covariance=cov, covariance_type=covariance_type)
    else:
        mean, conc = vmf_fit(flat, masked_flat, min_concentration, max_concentration)
        model.update(mean=mean, concentration=conc)
    eigvec, eigval = oc.cacg_m_step(np.swapaxes(yn[..., None, :, :], -1, -2), masked, q,
                                    hermitize=hermitize, covariance_norm=covariance_norm,
                                    eigenvalue_floor=eigenvalue_floor)
    model.update(eigvec=eigvec, eigval=eigval)
    return model


def joint_fit(kind, observation, embedding, initialization, iterations=100, saliency=None,
              affiliation_eps=1e-10, inline_permutation_alignment=False, **kw):
    """GCACGMMTrainer.fit gcacgmm.py:131-246 (kind='gaussian') /
    VMFCACGMMTrainer.fit vmfcacgmm.py:101-205 (kind='vmf')."""
    yn = unit_rows(observation)
    # NB: the embedding is NOT normalised here, also for kind='vmf': the reference's M-step
    # (vmfcacgmm.py:267-276) sees the raw rows, only VonMisesFisher.log_pdf normalises.
    if saliency is None:
        saliency = np.ones_like(initialization[..., 0, :])
    q = np.ones_like(initialization)
    aff = initialization
    model = None
    for _ in range(iterations):
        if model is not None:
            aff, q = joint_predict(model, yn, embedding, affiliation_eps,
                                   inline_permutation_alignment)
        model = joint_m_step(kind, yn, embedding, q, aff, saliency, **kw)
    return model


def joint_model_predict(model, observation, embedding):
    """GCACGMM.predict gcacgmm.py:47-64 / VMFCACGMM.predict vmfcacgmm.py:43-55."""
    yn = unit_rows(observation)
    if model['kind'] == 'vmf':
        embedding = unit_rows(embedding)
    return joint_predict(model, yn, embedding)[0]
