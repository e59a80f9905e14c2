"""Mask-based beamformer extraction on the device.

Mirrors pb_bss/extraction/beamformer.py (and pb_bss/math/solve.py for
`stable_solve`): same function names, argument meaning, shape conventions
(time last: X (..., D, T), mask (..., K, T), PSD (..., K, D, D)) and error
behaviour.  Bodies call the HIP library; NumPy in -> NumPy out, torch CUDA in
-> torch CUDA out.  Results are complex128 (float64 arithmetic on the device).
"""
import numpy as np

from .. import _lib, engine

__all__ = [
    'get_power_spectral_density_matrix', 'get_mvdr_vector_souden',
    'get_mvdr_vector', 'get_pca_vector', 'get_gev_vector',
    'blind_analytic_normalization', 'apply_beamforming_vector',
    'get_optimal_reference_channel', 'stable_solve', 'get_wmwf_vector',
    'get_pca', 'get_mvdr_vector_merl', 'get_lcmv_vector', 'get_lcmv_vector_souden',
    'distortionless_normalization', 'mvdr_snr_postfilter', 'zero_degree_normalization',
    'phase_correction', 'condition_covariance', 'apply_online_beamforming_vector',
]


def _res(x, like_torch):
    return x if like_torch else _lib.to_host(x)


def _c128(x):
    t = _lib.torch()
    return _lib.to_device(np.asarray(x) if not _lib.is_torch(x) else x, t.complex128)


def _obs(x):
    t = _lib.torch()
    x = _lib.to_device(x)
    if x.dtype not in (t.complex64, t.complex128):
        x = x.to(t.complex128)
    return x


def get_power_spectral_density_matrix(observation, mask=None, sensor_dim=-2,
                                      source_dim=-2, time_dim=-1, normalize=True):
    """Weighted PSD (covariance) matrices.  Reference: beamformer.py:59-160.

    observation (..., sensors, frames) complex; mask (bins, frames) /
    (..., frames) or (..., sources, frames); the *_dim arguments permute axes as
    in the reference.  Returns (..., sensors, sensors), (..., sources, sensors,
    sensors) or (sources, ..., sensors, sensors) for source_dim < -2.
    """
    like_torch = _lib.is_torch(observation)
    t = _lib.torch()
    x = _obs(observation)
    nd = x.ndim
    sensor_dim, source_dim, time_dim = (d % nd - nd for d in (sensor_dim, source_dim, time_dim))
    perm = [i for i in range(-nd, 0) if i not in (sensor_dim, time_dim)] + [sensor_dim, time_dim]
    x = x.permute(*[p % nd for p in perm])
    *lead, D, T = x.shape
    xb = x.reshape(-1, D, T).contiguous()
    if mask is None:
        out = engine.psd(xb, None)
        return _res(out.reshape(*lead, D, D), like_torch)
    m = _lib.to_device(mask)
    m = m.to(t.float64)  # bool / float32 masks are widened (reference :124-125)
    if m.ndim + 1 == nd:
        mb = m.expand(*lead, T).reshape(-1, 1, T).contiguous()
        out = engine.psd(xb, mb, normalize=normalize)
        return _res(out.reshape(*lead, D, D), like_torch)
    mperm = [i for i in range(-nd, 0) if i not in (source_dim, time_dim)] + [source_dim, time_dim]
    m = m.permute(*[p % nd for p in mperm])
    K = m.shape[-2]
    mb = m.expand(*lead, K, T).reshape(-1, K, T).contiguous()
    out = engine.psd(xb, mb, normalize=normalize).reshape(*lead, K, D, D)
    if source_dim < -2:
        # PSD shape (sources, ..., sensors, sensors) as the reference (:155-158)
        out = out.movedim(-3, source_dim % nd)
    return _res(out, like_torch)


def get_pca_vector(target_psd_matrix, scaling=None):
    """Principal eigenvector of the target PSD.  Reference: beamformer.py:163-224."""
    like_torch = _lib.is_torch(target_psd_matrix)
    t = _lib.torch()
    a = _c128(target_psd_matrix)
    *lead, D, _ = a.shape
    val, vec, st = engine.heev(a.reshape(-1, D, D).contiguous())
    if int(st.max().item()) != 0:
        raise np.linalg.LinAlgError('Eigenvalues did not converge')
    w = vec[:, :, -1].reshape(*lead, D)
    lam = val[:, -1].reshape(*lead)
    if scaling is None:
        pass
    elif scaling == 'trace':
        tr = t.einsum('...dd', a)
        w = w * (t.sqrt(tr) / t.linalg.vector_norm(w, dim=-1))[..., None]
    elif scaling == 'eigenvalue':
        w = w * (lam / t.linalg.vector_norm(w, dim=-1))[..., None]
    else:
        raise ValueError
    return _res(w, like_torch)


def stable_solve(A, B):
    """Batched A X = B with a minimum-norm least-squares answer for exactly
    singular matrices.  Reference: math/solve.py:20-114 (np.linalg.solve with
    per-matrix lstsq fallback); both branches run on the device."""
    like_torch = _lib.is_torch(A)
    a = _c128(A)
    b = _c128(B)
    assert a.shape[:-2] == b.shape[:-2], (a.shape, b.shape)
    assert a.shape[-1] == b.shape[-2], (a.shape, b.shape)
    D, M = b.shape[-2:]
    x, _ = engine.solve(a.reshape(-1, D, D).contiguous(), b.reshape(-1, D, M).contiguous())
    return _res(x.reshape(b.shape), like_torch)


def get_mvdr_vector(atf_vector, noise_psd_matrix):
    """w = Phi_nn^-1 h / (h^H Phi_nn^-1 h).  Reference: beamformer.py:230-260.
    atf_vector (..., bins, sensors); noise_psd_matrix (bins, sensors, sensors)."""
    assert noise_psd_matrix is not None
    like_torch = _lib.is_torch(atf_vector)
    h = _c128(atf_vector)
    n = _c128(noise_psd_matrix)
    while h.ndim > n.ndim - 1:
        n = n[None]
    n = n.expand(*h.shape[:-1], *n.shape[-2:])
    D = h.shape[-1]
    w, _ = engine.mvdr(h.reshape(-1, D).contiguous(), n.reshape(-1, D, D).contiguous())
    return _res(w.reshape(h.shape), like_torch)


def get_gev_vector(target_psd_matrix, noise_psd_matrix, force_cython=False,
                   use_eig=False):
    """GEV beamforming vector.  Reference: beamformer.py:292-364.

    use_eig=False (default): principal generalised eigenvector of the Hermitian-definite
    pencil, normalised like LAPACK zhegvd (w^H Phi_nn w = 1) -- cythonized/
    get_gev_vector.pyx:42-150 / the `eigh` fallback :367-411.  A noise PSD that is not
    positive definite raises ValueError under force_cython (the .pyx message) and
    numpy.linalg.LinAlgError otherwise (what the reference's SciPy fallback ends with).

    use_eig=True: the `eig` path (c_eig.pyx:14-123 zggev / scipy.linalg.eig, :352-358,
    :395-410): NO Hermitian or definiteness assumption, the eigenvalue numpy.argmax picks among
    the complex eigenvalues, UNIT-2-NORM eigenvector (the reference's normalisation on this
    path).  Like the reference it "crashes less often": only an exactly singular noise matrix
    or a failed QR iteration raise.
    """
    assert noise_psd_matrix is not None
    like_torch = _lib.is_torch(target_psd_matrix)
    a = _c128(target_psd_matrix)
    b = _c128(noise_psd_matrix)
    D = a.shape[-1]
    assert D == a.shape[-2], a.shape
    assert a.shape == b.shape, (a.shape, b.shape)
    if use_eig:
        w, _, st = engine.gev_general(a.reshape(-1, D, D).contiguous(),
                                      b.reshape(-1, D, D).contiguous())
        # one reduce + one read-back in the common case; the index search only on a failure
        # (`nonzero` is five launches and a second synchronisation of its own)
        if bool(st.any()):
            bad = (st != 0).nonzero()
            f = int(bad[0].item())
            code = int(st[f].item())
            if code & _lib.ST_EIG_NOCONV:
                msg = ('The QZ iteration failed.  No eigenvectors have been calculated '
                       f'for frequency {f}')  # c_eig.pyx:104-108
                if force_cython:
                    raise ValueError(msg)
            elif code & _lib.ST_SINGULAR:
                msg = f'noise PSD matrix of frequency {f} is exactly singular'
            else:
                msg = f'non-finite eigenvalue for frequency {f}'
            raise np.linalg.LinAlgError(f'Error for frequency {f}\n{msg}')  # :403-407
        return _res(w.reshape(a.shape[:-1]), like_torch)
    w, st = engine.gev(a.reshape(-1, D, D).contiguous(), b.reshape(-1, D, D).contiguous())
    if bool(st.any()):
        bad = (st != 0).nonzero()
        f = int(bad[0].item())
        code = int(st[f].item())
        if code & _lib.ST_NOT_POSDEF:
            msg = (f'the leading minor of order {code >> 8} of B is not positive '
                   'definite. The factorization of B could not be completed and '
                   f'no eigenvalues or eigenvectors were computed for frequency {f}')
        else:
            msg = f'Algorithm failed to compute an eigenvalue for frequency {f}'
        if force_cython:
            raise ValueError(msg)
        raise np.linalg.LinAlgError(f'Error for frequency {f}\n{msg}')
    return _res(w.reshape(a.shape[:-1]), like_torch)


def blind_analytic_normalization(vector, noise_psd_matrix):
    """BAN post-filter.  Reference: beamformer.py:459-488."""
    like_torch = _lib.is_torch(vector)
    w = _c128(vector)
    n = _c128(noise_psd_matrix)
    D = w.shape[-1]
    lead = np.broadcast_shapes(tuple(w.shape[:-1]), tuple(n.shape[:-2]))
    wb = w.expand(*lead, D).reshape(-1, D).contiguous()
    nb = n.expand(*lead, D, D).reshape(-1, D, D).contiguous()
    out = engine.ban(wb, nb)
    return _res(out.reshape(*lead, D), like_torch)


def apply_beamforming_vector(vector, mix):
    """s[..., t] = sum_d conj(w[..., d]) x[..., d, t].  Reference: beamformer.py:572-583."""
    like_torch = _lib.is_torch(mix)
    w = _c128(vector)
    x = _obs(mix)
    assert w.shape[-1] < 30, (w.shape, x.shape)
    D, T = x.shape[-2:]
    lead = np.broadcast_shapes(tuple(w.shape[:-1]), tuple(x.shape[:-2]))
    wb = w.expand(*lead, D).reshape(-1, D).contiguous()
    xl = tuple(x.shape[:-2])
    if len(xl) < len(lead) and lead[len(lead) - len(xl):] == xl:
        # the observation is shared along leading axes of `vector` (K beamformers per bin on one
        # STFT): the kernel reads x[b % Bx] instead of K expanded copies
        xb = x.reshape(-1, D, T).contiguous()
    else:
        xb = x.expand(*lead, D, T).reshape(-1, D, T).contiguous()
    out = engine.apply_bf(wb, xb)
    return _res(out.reshape(*lead, T), like_torch)


def _select_reference_channel(num, den, eps):
    """argmax_r sum_f num[f, r] / max(sum_f den[f, r], eps) with NumPy's complex
    ordering in the maximum (reference :616-624).  num/den: host (F, D)."""
    snr = num.sum(axis=0) / np.maximum(den.sum(axis=0), eps)
    assert np.all(np.isfinite(snr)), snr
    return int(np.argmax(snr.real))


def _select_reference_channel_device(num, den, eps):
    """The selection of `_select_reference_channel` without leaving the device (round 6: the
    chain of `pipeline.separate` / bench configs[3] used to stop for a device-to-host copy of
    2 x D numbers per problem in the middle of every step -- the host could not enqueue the rest
    of the step, nor the next fit, while it waited).  num / den: (..., F, D) device tensors.
    -> (int64 tensor (...) of argmax_r Re(sum_f num / max(sum_f den, eps)) -- first maximum, as
    np.argmax --, bool tensor (): every SNR finite -- what the reference asserts, :619; the
    caller checks it when it next synchronises (`pipeline.device_ops.assert_finite`))."""
    t = _lib.torch()
    n, d = num.sum(dim=-2), den.sum(dim=-2)
    # np.maximum on complex numbers orders by the real part first, then the imaginary part; the
    # denominators w^H N w are real up to rounding: compare the real parts, keep the complex value
    floor = t.full_like(d.real, eps)
    dd = t.where((d.real > floor) | ((d.real == floor) & (d.imag >= 0)), d, floor.to(d.dtype))
    snr = n / dd
    ok = t.isfinite(snr.real).all() & t.isfinite(snr.imag).all()
    return t.argmax(snr.real, dim=-1), ok


def _select_reference_channel_sharded(num, den, eps, shard_group):
    """The same selection when the frequency bins are sharded over the ranks of a
    torch.distributed group: the SNR of :616-620 sums over ALL bins, so
    the local sums -- 2 x D complex numbers per problem -- are all-reduced before the division.
    num/den: (..., F_local, D) tensors; returns an int array of shape (...)."""
    from ..sharding import all_reduce_sum
    t = _lib.torch()
    s = t.stack([num.sum(dim=-2), den.sum(dim=-2)])          # (2, ..., D)
    s = all_reduce_sum(s, None if shard_group is True else shard_group)
    s = s.cpu().numpy() if s.is_cuda else s.numpy()
    snr = s[0] / np.maximum(s[1], eps)
    assert np.all(np.isfinite(snr)), snr
    return np.argmax(snr.real, axis=-1)


def get_optimal_reference_channel(w_mat, target_psd_matrix, noise_psd_matrix, eps=None):
    """Reference channel maximising the post-filter SNR summed over ALL
    frequencies.  Reference: beamformer.py:601-624."""
    t = _lib.torch()
    w = _c128(w_mat)
    if w.ndim != 3:
        raise ValueError(
            'Estimating the ref_channel expects currently that the input '
            'has 3 ndims (frequency x sensors x sensors). '
            'Considering an independent dim in the SNR estimate is not '
            'unique.')
    if eps is None:
        eps = np.finfo(np.float64).tiny
    tp = _c128(target_psd_matrix).expand(w.shape).contiguous()
    nn = _c128(noise_psd_matrix).expand(w.shape).contiguous()
    num, den = engine.reference_channel_terms(w.contiguous(), tp, nn)
    return _select_reference_channel(_lib.to_host(num), _lib.to_host(den), eps)


def get_mvdr_vector_souden(target_psd_matrix, noise_psd_matrix, ref_channel=None,
                           eps=None, return_ref_channel=False, *, shard_group=None):
    """MVDR beamformer in the Souden formulation.  Reference:
    beamformer.py:627-698.  (..., bins, sensors, sensors) -> (..., bins, sensors).
    The automatic reference channel needs exactly 3 dims (bins first) because
    its SNR estimate sums over all frequencies.

    shard_group (extension): the bins axis holds only THIS RANK'S block of frequency bins of a
    torch.distributed group (True = the default group); the automatic reference channel is then
    chosen from the SNR summed over the bins of all ranks (one all-reduce of 2 x D numbers)."""
    assert noise_psd_matrix is not None
    like_torch = _lib.is_torch(target_psd_matrix)
    tp = _c128(target_psd_matrix)
    nn = _c128(noise_psd_matrix)
    D = tp.shape[-1]
    if eps is None:
        eps = np.finfo(np.float64).tiny
    mat, num, den, _ = engine.mvdr_souden(
        tp.reshape(-1, D, D).contiguous(),
        nn.expand(tp.shape).reshape(-1, D, D).contiguous(), eps)
    if ref_channel is None:
        if tp.ndim != 3:
            raise ValueError(
                'Estimating the ref_channel expects currently that the input '
                'has 3 ndims (frequency x sensors x sensors). '
                'Considering an independent dim in the SNR estimate is not '
                'unique.')
        if shard_group is not None:
            ref_channel = int(_select_reference_channel_sharded(num, den, eps, shard_group))
        else:
            ref_channel = _select_reference_channel(_lib.to_host(num), _lib.to_host(den), eps)
    assert np.isscalar(ref_channel), ref_channel
    w = mat.reshape(*tp.shape)[..., ref_channel]
    if return_ref_channel:
        return _res(w, like_torch), ref_channel
    return _res(w, like_torch)


def get_wmwf_vector(target_psd_matrix, noise_psd_matrix, reference_channel=None,
                    channel_selection_vector=None, distortion_weight=1.):
    """Speech-distortion-weighted multichannel Wiener filter.  Reference:
    beamformer.py:701-753.  filter = (Phi_nn^-1 Phi_xx) / (mu + trace) (or
    / sqrt(Phi_xx[0,0] * trace) for distortion_weight='frequency_dependent');
    the result is the reference-channel column, a `channel_selection_vector`
    weighted sum of columns, or the SNR-optimal column when neither is given."""
    assert noise_psd_matrix is not None
    like_torch = _lib.is_torch(target_psd_matrix)
    tp = _c128(target_psd_matrix)
    nn = _c128(noise_psd_matrix)
    D = tp.shape[-1]
    freq_dep = isinstance(distortion_weight, str)
    if freq_dep:
        assert distortion_weight == 'frequency_dependent', distortion_weight
    mat, num, den, _ = engine.wmwf(
        tp.reshape(-1, D, D).contiguous(), nn.expand(tp.shape).reshape(-1, D, D).contiguous(),
        0.0 if freq_dep else float(distortion_weight), freq_dep)
    filt = mat.reshape(*tp.shape)
    if channel_selection_vector is not None:
        sel = _lib.to_device(channel_selection_vector).to(filt.device).to(filt.dtype)
        return _res((filt * sel[..., None, :]).sum(dim=-1), like_torch)
    if reference_channel is None:
        if tp.ndim != 3:
            raise ValueError(
                'Estimating the ref_channel expects currently that the input '
                'has 3 ndims (frequency x sensors x sensors). '
                'Considering an independent dim in the SNR estimate is not '
                'unique.')
        reference_channel = _select_reference_channel(
            _lib.to_host(num), _lib.to_host(den), np.finfo(np.float64).tiny)
    assert np.isscalar(reference_channel), reference_channel
    return _res(filt[..., reference_channel], like_torch)


# ---- remaining members of the family -----------------------
def get_pca(target_psd_matrix, return_all_vecs=False):
    """All principal components and eigenvalues, or the dominant pair.
    Reference: beamformer.py:163-194."""
    like_torch = _lib.is_torch(target_psd_matrix)
    a = _c128(target_psd_matrix)
    *lead, D, _ = a.shape
    val, vec, st = engine.heev(a.reshape(-1, D, D).contiguous())
    if int(st.max().item()) != 0:
        raise np.linalg.LinAlgError('Eigenvalues did not converge')
    if return_all_vecs:
        return _res(vec.reshape(*lead, D, D), like_torch), _res(val.reshape(*lead, D), like_torch)
    return (_res(vec[:, :, -1].reshape(*lead, D), like_torch),
            _res(val[:, -1].reshape(*lead), like_torch))


def get_mvdr_vector_merl(target_psd_matrix, noise_psd_matrix):
    """MVDR variant of MERL TR2016-072: h = G / tr G with G = Phi_nn^-1 Phi_xx, reference
    channel maximising the post-SNR summed over frequency.  Reference: beamformer.py:263-289.
    The filter is the mu = 0 case of the wMWF kernel (complex trace, no floor)."""
    like_torch = _lib.is_torch(target_psd_matrix)
    tp = _c128(target_psd_matrix)
    nn = _c128(noise_psd_matrix)
    D = tp.shape[-1]
    mat, num, den, st = engine.wmwf(tp.reshape(-1, D, D).contiguous(),
                                    nn.expand(tp.shape).reshape(-1, D, D).contiguous(), 0.0, False)
    if int(st.max().item()) != 0:
        raise np.linalg.LinAlgError('Singular matrix')  # np.linalg.solve (:276)
    nom = _lib.to_host(num).sum(axis=0)
    denom = _lib.to_host(den).sum(axis=0)
    h_idx = int(np.argmax(nom / denom))
    return _res(mat.reshape(*tp.shape)[..., h_idx], like_torch)


def get_lcmv_vector(atf_vectors, response_vector, noise_psd_matrix):
    """LCMV beamformer.  atf_vectors (targets, bins, sensors); response_vector (targets,),
    e.g. [1, 0, ..., 0]; noise_psd_matrix (bins, sensors, sensors) -> (bins, sensors).
    Reference: beamformer.py:414-456."""
    like_torch = _lib.is_torch(atf_vectors)
    atf = _c128(atf_vectors).contiguous()
    K, F, D = atf.shape
    nn = _c128(noise_psd_matrix).contiguous()
    assert tuple(nn.shape) == (F, D, D), nn.shape
    resp = _c128(np.asarray(response_vector) if not _lib.is_torch(response_vector)
                 else response_vector).to(atf.device).contiguous()
    assert tuple(resp.shape) == (K,), resp.shape
    w, _ = engine.lcmv(atf, resp, nn)
    return _res(w, like_torch)


def get_lcmv_vector_souden(target_psd_matrix, interference_psd_matrix, noise_psd_matrix,
                           ref_channel=None, eps=None, return_ref_channel=False):
    """Not available, exactly as in the reference (beamformer.py:756-786)."""
    raise NotImplementedError(
        'This is not yet thoroughly tested. It also misses the response vector,'
        'thus it is unclear, how to select, which speaker to attend to.')


def distortionless_normalization(vector, atf_vector, noise_psd_matrix):
    """(Phi w w^H / (w^H Phi w)) a per bin.  Reference: beamformer.py:491-499."""
    like_torch = _lib.is_torch(vector)
    w = _c128(vector).contiguous()
    return _res(engine.distortionless_normalization(
        w, _c128(atf_vector).to(w.device).contiguous(),
        _c128(noise_psd_matrix).to(w.device).contiguous()), like_torch)


def mvdr_snr_postfilter(vector, target_psd_matrix, noise_psd_matrix):
    """(w^H Phi_xx w) / (w^H Phi_nn w), shape (bins, 1).  Reference: beamformer.py:502-509."""
    like_torch = _lib.is_torch(vector)
    w = _c128(vector).contiguous()
    out = engine.snr_postfilter(w, _c128(target_psd_matrix).to(w.device).contiguous(),
                                _c128(noise_psd_matrix).to(w.device).contiguous())
    return _res(out[:, None], like_torch)


def zero_degree_normalization(vector, reference_channel):
    """Rotate every vector so that its reference channel is real and non-negative.
    Reference: beamformer.py:512-514."""
    like_torch = _lib.is_torch(vector)
    v = _c128(vector)
    D = v.shape[-1]
    out = engine.zero_degree_normalization(v.reshape(-1, D).contiguous(), reference_channel)
    return _res(out.reshape(v.shape), like_torch)


def phase_correction(vector):
    """Phase correction to reduce distortions due to phase inconsistencies between
    neighbouring bins.  vector (..., bins, sensors).  Reference: beamformer.py:517-560 (the
    vectorised form, whose running product follows axis 0 of the phasor array)."""
    like_torch = _lib.is_torch(vector)
    v = _c128(vector).contiguous()
    assert v.ndim >= 2, v.shape
    return _res(engine.phase_correction(v), like_torch)


def condition_covariance(x, gamma):
    """(x + gamma tr(x)/D I) / (1 + gamma).  Reference: beamformer.py:563-569."""
    like_torch = _lib.is_torch(x)
    a = _c128(x)
    D = a.shape[-1]
    out = engine.condition_covariance(a.reshape(-1, D, D).contiguous(), gamma)
    return _res(out.reshape(a.shape), like_torch)


def apply_online_beamforming_vector(vector, mix):
    """Time-dependent beamforming vectors: vector (frames, bins, sensors), mix
    (bins, sensors, frames) -> (bins, frames).  Reference: beamformer.py:586-598."""
    like_torch = _lib.is_torch(mix)
    x = _obs(mix).contiguous()
    v = _c128(vector).to(x.device).contiguous()
    assert v.ndim == 3 and x.ndim == 3, (v.shape, x.shape)
    T, F, D = v.shape
    assert tuple(x.shape) == (F, D, T), (v.shape, x.shape)
    return _res(engine.apply_online_bf(v, x), like_torch)
