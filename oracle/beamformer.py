"""NumPy/SciPy restatement of the reference beamformer path (TEST INFRASTRUCTURE).

Oracle for pb_bss.extraction.beamformer / beamformer_wrapper / math.solve.
Never imported by the product package.  Citations are ``file:line`` under
/root/reference/pb_bss/.  Pinned against the real reference by
oracle/make_golden.py -> tests/golden/.
"""
import numpy as np
import scipy.linalg

__all__ = [
    'psd', 'gev_vector', 'stable_solve', 'optimal_reference_channel',
    'mvdr_souden', 'mvdr', 'ban', 'apply_bf', 'pca_vector', 'bf_vector', 'wmwf',
    'pca', 'mvdr_merl', 'lcmv', 'distortionless_normalization', 'mvdr_snr_postfilter',
    'zero_degree_normalization', 'phase_correction', 'condition_covariance', 'apply_online_bf',
]


def psd(observation, mask=None, normalize=True):
    """extraction/beamformer.py:59-160 for the default axis arguments.
    observation (..., D, T); mask (..., K, T) or (..., T) -> (..., [K,] D, D)."""
    if mask is None:
        p = np.einsum('...dt,...et->...de', observation, observation.conj())
        return p / observation.shape[-1]
    mask = np.array(mask, dtype=np.float64 if np.asarray(mask).dtype == bool
                    else np.asarray(mask).dtype, copy=True)
    if normalize:
        mask = mask / np.maximum(np.sum(mask, axis=-1, keepdims=True), 1e-10)
    if mask.ndim + 1 == observation.ndim:
        return np.einsum('...dt,...et->...de', mask[..., None, :] * observation,
                         observation.conj())
    return np.einsum('...kt,...dt,...et->...kde', mask, observation,
                     observation.conj())


def gev_vector(target_psd, noise_psd):
    """extraction/beamformer.py:367-411 (_get_gev_vector, eigh branch) ==
    cythonized/get_gev_vector.pyx:42-150 up to the eigenvector phase:
    principal generalised eigenvector, normalised w^H Phi_nn w = 1."""
    D = target_psd.shape[-1]
    shape = target_psd.shape
    t = target_psd.reshape(-1, D, D)
    n = noise_psd.reshape(-1, D, D)
    out = np.empty((t.shape[0], D), dtype=np.complex128)
    for f in range(t.shape[0]):
        vals, vecs = scipy.linalg.eigh(t[f], n[f])
        out[f] = vecs[:, np.argmax(vals)]
    return out.reshape(shape[:-1])


def gev_vector_eig(target_psd, noise_psd, return_eigenvalue=False):
    """`get_gev_vector(..., use_eig=True)`: extraction/beamformer.py:352-358 ->
    cythonized/c_eig.pyx:14-123 (zggev, eigenvalues alpha/beta, eigenvectors renormalised to unit
    2-norm :121) or, without the compiled module, :367-411 with scipy.linalg.eig (also unit
    2-norm).  Both pick numpy.argmax of the COMPLEX eigenvalues (real part first, then imaginary)
    and make no Hermitian / definiteness assumption.  Phase of the vector: arbitrary."""
    D = target_psd.shape[-1]
    shape = target_psd.shape
    t = target_psd.reshape(-1, D, D)
    n = noise_psd.reshape(-1, D, D)
    out = np.empty((t.shape[0], D), dtype=np.complex128)
    lam = np.empty(t.shape[0], dtype=np.complex128)
    for f in range(t.shape[0]):
        vals, vecs = scipy.linalg.eig(t[f], n[f])
        k = int(np.argmax(vals))
        out[f] = vecs[:, k] / np.linalg.norm(vecs[:, k])
        lam[f] = vals[k]
    out = out.reshape(shape[:-1])
    return (out, lam.reshape(shape[:-2])) if return_eigenvalue else out


def stable_solve(A, B):
    """math/solve.py:20-114: batched solve, per-matrix lstsq on singular ones."""
    A = np.asarray(A)
    B = np.asarray(B)
    try:
        return np.linalg.solve(A, B)
    except np.linalg.LinAlgError:
        a = A.reshape(-1, *A.shape[-2:])
        b = B.reshape(-1, *B.shape[-2:])
        c = np.zeros_like(b)
        for i in range(a.shape[0]):
            try:
                c[i] = np.linalg.solve(a[i], b[i])
            except np.linalg.LinAlgError:
                c[i] = np.linalg.lstsq(a[i], b[i], rcond=None)[0]
        return c.reshape(B.shape)


def optimal_reference_channel(w_mat, target_psd, noise_psd, eps=None):
    """extraction/beamformer.py:601-624."""
    if w_mat.ndim != 3:
        raise ValueError('expects (frequency, sensors, sensors)')
    if eps is None:
        eps = np.finfo(w_mat.dtype).tiny
    num = np.einsum('...FdR,...FdD,...FDR->...R', w_mat.conj(), target_psd, w_mat)
    den = np.einsum('...FdR,...FdD,...FDR->...R', w_mat.conj(), noise_psd, w_mat)
    snr = num / np.maximum(den, eps)
    assert np.all(np.isfinite(snr)), snr
    return np.argmax(snr.real)


def mvdr_souden(target_psd, noise_psd, ref_channel=None, eps=None,
                return_ref_channel=False):
    """extraction/beamformer.py:627-698."""
    phi = stable_solve(noise_psd, target_psd)
    lam = np.trace(phi, axis1=-1, axis2=-2)[..., None, None]
    if eps is None:
        eps = np.finfo(lam.dtype).tiny
    mat = phi / np.maximum(lam.real, eps)
    if ref_channel is None:
        ref_channel = optimal_reference_channel(mat, target_psd, noise_psd, eps=eps)
    w = mat[..., ref_channel]
    return (w, ref_channel) if return_ref_channel else w


def mvdr(atf_vector, noise_psd):
    """extraction/beamformer.py:230-260."""
    while atf_vector.ndim > noise_psd.ndim - 1:
        noise_psd = noise_psd[None]
    noise_psd = 0.5 * (noise_psd + np.conj(noise_psd.swapaxes(-1, -2)))
    num = np.linalg.solve(noise_psd, atf_vector[..., None])[..., 0]
    den = np.einsum('...d,...d->...', atf_vector.conj(), num)
    return num / den[..., None]


def ban(vector, noise_psd):
    """extraction/beamformer.py:459-488."""
    nom = np.sqrt(np.einsum('...a,...ab,...bc,...c->...', vector.conj(),
                            noise_psd, noise_psd, vector))
    den = np.einsum('...a,...ab,...b->...', vector.conj(), noise_psd, vector)
    den = np.sqrt(den * den.conj())
    norm = np.divide(nom, den, out=np.zeros_like(nom), where=den != 0)
    return vector * np.abs(norm[..., None])


def wmwf(target_psd, noise_psd, reference_channel=None, channel_selection_vector=None,
         distortion_weight=1.):
    """extraction/beamformer.py:701-753."""
    phi = stable_solve(noise_psd, target_psd)
    lam = np.trace(phi, axis1=-1, axis2=-2)[..., None, None]
    if isinstance(distortion_weight, str):
        filt = phi / np.sqrt(target_psd[..., 0:1, 0:1] * lam)
    else:
        filt = phi / (distortion_weight + lam)
    if channel_selection_vector is not None:
        return np.sum(filt * channel_selection_vector[..., None, :], axis=-1)
    if reference_channel is None:
        reference_channel = optimal_reference_channel(filt, target_psd, noise_psd)
    return filt[..., reference_channel]


def apply_bf(vector, mix):
    """extraction/beamformer.py:572-583."""
    assert vector.shape[-1] < 30
    return np.einsum('...a,...at->...t', vector.conj(), mix)


def pca_vector(target_psd):
    """extraction/beamformer.py:163-224 (scaling=None)."""
    shape = target_psd.shape
    vals, vecs = np.linalg.eigh(target_psd.reshape(-1, *shape[-2:]))
    return vecs[..., -1].reshape(shape[:-1])


def rank_one_estimate(kind, cov, noise_psd=None):
    """extraction/beamformer_wrapper.py:11-25 ('rank1_pca': get_pca_rank_one_estimate) and :49-69
    ('rank1_gev': get_gev_rank_one_estimate): a a^H scaled to the trace of `cov`, with a the
    dominant eigenvector or the GEV ATF estimate Phi_nn w_gev (:28-47)."""
    if kind == 'rank1_pca':
        a = pca_vector(cov)
    elif kind == 'rank1_gev':
        a = np.einsum('...dD,...D->...d', noise_psd, gev_vector(cov, noise_psd))
    else:
        raise ValueError(kind)
    r1 = np.einsum('...d,...D->...dD', a, a.conj())
    scale = np.trace(cov, axis1=-1, axis2=-2) / np.trace(r1, axis1=-1, axis2=-2)
    return scale[..., None, None] * r1


def bf_vector(beamformer, target_psd, noise_psd=None, **kw):
    """extraction/beamformer_wrapper.py:117-236 for the cores on the hot path:
    'gev', 'mvdr_souden', 'pca', 'pca+mvdr', 'scaled_gev_atf+mvdr',
    'rank1_gev+...' / 'rank1_pca+...', 'ch<N>', each optionally with '+ban'."""
    do_ban = beamformer.endswith('+ban')
    core = beamformer[:-4] if do_ban else beamformer

    def rank1(kind, cov):
        return rank_one_estimate(kind, cov, noise_psd)

    if core == 'pca':
        w = pca_vector(target_psd)
    elif core in ('pca+mvdr', 'scaled_gev_atf+mvdr'):
        if core.startswith('pca'):
            atf = pca_vector(target_psd)
        else:
            atf = np.einsum('...dD,...D->...d', noise_psd,
                            gev_vector(target_psd, noise_psd))
        w = mvdr(atf, noise_psd)
    elif core in ('mvdr_souden', 'rank1_pca+mvdr_souden', 'rank1_gev+mvdr_souden'):
        if core != 'mvdr_souden':
            target_psd = rank1(core.split('+')[0], target_psd)
        w = mvdr_souden(target_psd, noise_psd, **kw)
    elif core in ('wmwf', 'rank1_pca+wmwf', 'rank1_gev+wmwf'):
        if core != 'wmwf':
            target_psd = rank1(core.split('+')[0], target_psd)
        w = wmwf(target_psd, noise_psd, **kw)
    elif core in ('gev', 'rank1_pca+gev', 'rank1_gev+gev'):
        if core != 'gev':
            target_psd = rank1(core.split('+')[0], target_psd)
        w = gev_vector(target_psd, noise_psd)
    elif core.startswith('ch') and core[2:].isdigit():
        D = target_psd.shape[-1]
        w = np.zeros(D)
        w[int(core[2:])] = 1
        w = np.broadcast_to(w, target_psd.shape[:-1])
    else:
        raise ValueError(f'Could not find implementation for {core}.')
    if do_ban:
        w = ban(w, noise_psd)
    return w


# ---- remaining members of the family -----------------------
def pca(target_psd, return_all_vecs=False):
    """extraction/beamformer.py:163-194."""
    shape = target_psd.shape
    vals, vecs = np.linalg.eigh(target_psd.reshape(-1, *shape[-2:]))
    if return_all_vecs:
        return vecs.reshape(shape), vals.reshape(shape[:-1])
    return vecs[..., -1].reshape(shape[:-1]), vals[..., -1].reshape(shape[:-2])


def mvdr_merl(target_psd, noise_psd):
    """extraction/beamformer.py:263-289."""
    G = np.linalg.solve(noise_psd, target_psd)
    h = G / np.trace(G, axis1=-2, axis2=-1)[..., None, None]
    nom = np.sum(np.einsum('...fac,fab,...fbc->c', h.conj(), target_psd, h))
    den = np.sum(np.einsum('...fac,fab,...fbc->c', h.conj(), noise_psd, h))
    # NB: np.sum over the (sensors,) result gives scalars in the reference, so nom/denom is
    # a scalar and argmax is 0 -- restated as written (beamformer.py:281-289).
    return h[..., int(np.argmax(nom / den))]


def lcmv(atf_vectors, response_vector, noise_psd):
    """extraction/beamformer.py:414-456."""
    response_vector = np.asarray(response_vector)
    K, F, D = atf_vectors.shape
    pih = stable_solve(np.broadcast_to(noise_psd[None], (K, F, D, D)),
                       atf_vectors[..., None])[..., 0]                     # (K,F,D)
    hph = np.einsum('k...d,K...d->...kK', atf_vectors.conj(), pih)        # (F,K,K)
    rv = np.repeat(response_vector[None, :, None].astype(np.complex64), F, axis=0)
    temp = stable_solve(hph, rv)[..., 0]                                  # (F,K)
    return np.einsum('k...d,...k->...d', pih, temp)


def distortionless_normalization(vector, atf_vector, noise_psd):
    """extraction/beamformer.py:491-499."""
    nom = np.einsum('fab,fb,fc->fac', noise_psd, vector, vector.conj())
    den = np.einsum('fa,fab,fb->f', vector.conj(), noise_psd, vector)
    return np.einsum('fab,fb->fa', nom / den[..., None, None], atf_vector)


def mvdr_snr_postfilter(vector, target_psd, noise_psd):
    """extraction/beamformer.py:502-509."""
    nom = np.einsum('fa,fab,fb->f', vector.conj(), target_psd, vector)
    den = np.einsum('fa,fab,fb->f', vector.conj(), noise_psd, vector)
    return (nom / den)[:, None]


def zero_degree_normalization(vector, reference_channel):
    """extraction/beamformer.py:512-514."""
    return vector * np.exp(-1j * np.angle(vector[..., reference_channel, None]))


def phase_correction(vector):
    """extraction/beamformer.py:517-560 (the vectorised body: running product along
    axis 0 of the (..., F-1, 1) phasor array)."""
    vector = np.array(vector, copy=True)
    inner = np.sum(vector[..., 1:, :].conj() * vector[..., :-1, :], axis=-1, keepdims=True)
    vector[..., 1:, :] *= np.cumprod(np.exp(1j * np.angle(inner)), axis=0)
    return vector


def condition_covariance(x, gamma):
    """extraction/beamformer.py:563-569."""
    D = x.shape[-1]
    scale = gamma * np.trace(x, axis1=-2, axis2=-1) / D
    return (x + np.eye(D) * scale[..., None, None]) / (1 + gamma)


def apply_online_bf(vector, mix):
    """extraction/beamformer.py:586-598."""
    return np.einsum('...at,...at->...t', vector.transpose(1, 2, 0).conj(), mix)
