"""Thin device-tensor layer over the C ABI: every function takes/returns
contiguous torch CUDA tensors with the independent axes flattened to one batch
axis B, launches exactly the library call named in its docstring on the current
torch stream, and raises the reference's exception types from status words.

No arithmetic is done here beyond shape bookkeeping.
"""
import ctypes
import os

import numpy as np

from . import _lib


def _t():
    return _lib.torch()


def _status_bits(status):
    """OR of all status words (one small D2H copy; 0 without it when every word is 0)."""
    if not status.numel() or int(status.max().item()) == 0:
        return 0
    return int(np.bitwise_or.reduce(_lib.to_host(status).ravel()))


class StatusAssertionError(AssertionError):
    """The reference's `assert np.isfinite(...)` failure, derived from a kernel's status words
    (an AssertionError for callers that catch the reference's; a type of its own for callers that
    must tell a status failure from an argument check)."""


class StatusLinAlgError(np.linalg.LinAlgError):
    """The reference's eigensolver failure, derived from a kernel's status words."""


def _raise_for_bits(bits, what):
    """Mirror the reference's failure modes of the M-step."""
    if bits & _lib.ST_NONFINITE:
        # distribution/complex_angular_central_gaussian.py:127, :326, :333
        raise StatusAssertionError(
            f'{what}: non-finite covariance / eigenvalues '
            '(reference: assert np.isfinite(...))')
    if bits & _lib.ST_EIG_NOCONV:
        # complex_angular_central_gaussian.py:94-110 (LinAlgError from eigh/eig)
        raise StatusLinAlgError(f'{what}: Hermitian eigensolver did not converge')


def _status_raise_em(status, what):
    _raise_for_bits(_status_bits(status), what)


_SPLIT_POISON = _lib.ST_NONFINITE | _lib.ST_EIG_NOCONV


def _checked_with_split_retry(launch, dev, what):
    """Run launch() -> dict with 'status' and raise for its status bits.  The split groups of a
    remainder bin (2^n + 1 bins) exchange their sums through bounded waits; when one runs out
    -- the member workgroups were not on the chip together: compute units held by other work --
    the launch marks the problems concerned with NONFINITE | EIG_NOCONV and raises the flag of
    pbbss_split_error.  That is not a numerical failure: repeat once without split groups."""
    r = launch()
    bits = _status_bits(r['status'])
    if (bits & _SPLIT_POISON) == _SPLIT_POISON and split_error(dev.index):
        import warnings
        warnings.warn(f'{what}: the split groups of the remainder bin were not co-resident (GPU '
                      'shared with other work); repeating the fit without them', RuntimeWarning)
        # consume the report: counters and the sticky flag go back to zero, so that a later
        # genuine NONFINITE | EIG_NOCONV failure of this handle is raised, not refitted
        split_reset(dev.index)
        was = split_tail(dev.index)
        set_split_tail(False, dev.index)
        try:
            r = launch()
            bits = _status_bits(r['status'])
        finally:
            set_split_tail(was, dev.index)  # what the caller had chosen, not unconditionally on
    _raise_for_bits(bits, what)
    return r


def normalize_observation(y):
    """pbbss_normalize_observation: (B,T,D) complex -> (B,D,T), unit norm."""
    t = _t()
    B, T, D = y.shape
    is128 = y.dtype == t.complex128
    out = t.empty((B, D, T), dtype=y.dtype, device=y.device)
    rc = _lib.load().pbbss_normalize_observation(
        _lib.handle(y.device.index), _lib.ptr(y), int(is128), B, T, D,
        _lib.ptr(out), _lib.stream_ptr(y.device.index))
    _lib.check(rc, 'normalize_observation')
    return out


def em_fit(y, K, *, gamma0=None, model=None, iterations=100, saliency=None,
           activity=None, covariance_norm='eigenvalue', weight_mode=0,
           affiliation_eps=1e-10, eigenvalue_floor=1e-10, hermitize=True,
           layout=_lib.LAYOUT_TD, final_predict=False, return_q=False,
           force_eig=False, check_status=True, precision='f64'):
    """pbbss_cacgmm_fit.  y (B,T,D) [layout TD] or (B,D,T) [layout DT] complex.
    precision 'f32': the packed-FP32 "reference precision" kernel (complex64 y, D <= 8, K <= 6,
    LDS-resident frames) -- NotImplementedError where it does not apply.

    gamma0 (B,K,T) f64, or model=(eigvec (B,K,D,D) c128, eigval (B,K,D), weight (B,K)).
    Returns dict(eigvec, eigval, weight (B,K), status, affiliation?, quadratic_form?).
    """
    t = _t()
    dev = y.device
    if layout == _lib.LAYOUT_TD:
        B, T, D = y.shape
    else:
        B, D, T = y.shape
    is128 = y.dtype == t.complex128
    assert y.dtype in (t.complex64, t.complex128), y.dtype
    opts = _lib.EmOpts(
        iterations=int(iterations),
        covariance_norm=_lib.COVNORM[covariance_norm],
        weight_mode=int(weight_mode), hermitize=int(bool(hermitize)),
        layout=int(layout), y_is_c128=int(is128),
        final_predict=int(bool(final_predict)), force_eig=int(bool(force_eig)),
        affiliation_eps=float(affiliation_eps),
        eigenvalue_floor=float(eigenvalue_floor), precision=_lib.PRECISION[precision])
    f64 = t.float64
    if model is not None:
        in_vec, in_val, in_w = model
        assert in_vec.shape == (B, K, D, D) and in_val.shape == (B, K, D) and in_w.shape == (B, K)
    else:
        in_vec = in_val = in_w = None
        assert gamma0.shape == (B, K, T) and gamma0.dtype == f64, (gamma0.shape, gamma0.dtype)
    if saliency is not None:
        assert saliency.shape == (B, T) and saliency.dtype == f64
    if activity is not None:
        assert activity.shape == (B, K, T) and activity.dtype == t.uint8

    def launch():
        out_vec = t.empty((B, K, D, D), dtype=t.complex128, device=dev)
        out_val = t.empty((B, K, D), dtype=f64, device=dev)
        out_w = t.empty((B, K), dtype=f64, device=dev)
        out_st = t.empty((B, K), dtype=t.int32, device=dev)  # every row is written by the library
        out_aff = t.empty((B, K, T), dtype=f64, device=dev) if final_predict else None
        out_q = t.empty((B, K, T), dtype=f64, device=dev) if (final_predict and return_q) else None
        rc = _lib.load().pbbss_cacgmm_fit(
            _lib.handle(dev.index), _lib.ptr(y), B, T, D, K, _lib.ptr(gamma0),
            _lib.ptr(in_vec), _lib.ptr(in_val), _lib.ptr(in_w), _lib.ptr(saliency),
            _lib.ptr(activity), ctypes.byref(opts), _lib.ptr(out_vec),
            _lib.ptr(out_val), _lib.ptr(out_w), _lib.ptr(out_st), _lib.ptr(out_aff),
            _lib.ptr(out_q), _lib.stream_ptr(dev.index))
        _lib.check(rc, f'cacgmm_fit(B={B},T={T},D={D},K={K})')
        return dict(eigvec=out_vec, eigval=out_val, weight=out_w, status=out_st,
                    affiliation=out_aff, quadratic_form=out_q)

    if check_status:
        return _checked_with_split_retry(launch, dev, 'CACGMMTrainer.fit')
    return launch()


def em_fit_shared(y, K, group, *, weight_mode, gamma0=None, model=None, iterations=100,
                  saliency=None, activity=None, covariance_norm='eigenvalue',
                  affiliation_eps=1e-10, eigenvalue_floor=1e-10, final_predict=False,
                  return_q=False, force_eig=False, check_status=True):
    """pbbss_cacgmm_fit_shared: the EM loop with mixture weights estimated over `group`
    consecutive problems (weight_constant_axis (-3,) -> WEIGHT_SHARED_KT, (-3, -1) ->
    WEIGHT_SHARED_K), one cooperative launch.  y (B,T,D) complex, B = n_groups * group.
    model = (eigvec (B,K,D,D), eigval (B,K,D), weight (B/group, K[, T])).
    Returns None when the configuration is not served by the cooperative kernel
    (PBBSS_ERR_UNSUPPORTED: too many bins to be co-resident, K > 4, long utterances);
    otherwise dict(eigvec, eigval, weight (B/group, K[, T]), status, affiliation?, ...)."""
    t = _t()
    dev = y.device
    B, T, D = y.shape
    assert B % group == 0, (B, group)
    G = B // group
    is128 = y.dtype == t.complex128
    assert y.dtype in (t.complex64, t.complex128), y.dtype
    assert weight_mode in (_lib.WEIGHT_SHARED_K, _lib.WEIGHT_SHARED_KT), weight_mode
    wshape = (G, K) if weight_mode == _lib.WEIGHT_SHARED_K else (G, K, T)
    opts = _lib.EmOpts(
        iterations=int(iterations), covariance_norm=_lib.COVNORM[covariance_norm],
        weight_mode=int(weight_mode), hermitize=1, layout=int(_lib.LAYOUT_TD),
        y_is_c128=int(is128), final_predict=int(bool(final_predict)),
        force_eig=int(bool(force_eig)), affiliation_eps=float(affiliation_eps),
        eigenvalue_floor=float(eigenvalue_floor))
    f64 = t.float64
    out_vec = t.empty((B, K, D, D), dtype=t.complex128, device=dev)
    out_val = t.empty((B, K, D), dtype=f64, device=dev)
    out_w = t.empty(wshape, dtype=f64, device=dev)
    out_st = t.zeros((B, K), dtype=t.int32, device=dev)
    out_aff = t.empty((B, K, T), dtype=f64, device=dev) if final_predict else None
    out_q = t.empty((B, K, T), dtype=f64, device=dev) if (final_predict and return_q) else None
    if model is not None:
        in_vec, in_val, in_w = model
        assert in_vec.shape == (B, K, D, D) and in_val.shape == (B, K, D), in_vec.shape
        assert tuple(in_w.shape) == wshape and in_w.is_contiguous(), (in_w.shape, wshape)
    else:
        in_vec = in_val = in_w = None
        assert gamma0.shape == (B, K, T) and gamma0.dtype == f64, (gamma0.shape, gamma0.dtype)
    if saliency is not None:
        assert saliency.shape == (B, T) and saliency.dtype == f64
    if activity is not None:
        assert activity.shape == (B, K, T) and activity.dtype == t.uint8
    rc = _lib.load().pbbss_cacgmm_fit_shared(
        _lib.handle(dev.index), _lib.ptr(y), B, T, D, K, int(group), _lib.ptr(gamma0),
        _lib.ptr(in_vec), _lib.ptr(in_val), _lib.ptr(in_w), _lib.ptr(saliency),
        _lib.ptr(activity), ctypes.byref(opts), _lib.ptr(out_vec),
        _lib.ptr(out_val), _lib.ptr(out_w), _lib.ptr(out_st), _lib.ptr(out_aff),
        _lib.ptr(out_q), _lib.stream_ptr(dev.index))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, f'cacgmm_fit_shared(B={B},group={group},T={T},D={D},K={K})')
    if check_status:
        poison = _lib.ST_NONFINITE | _lib.ST_EIG_NOCONV
        # the status test first: split_error() is a blocking read of the handle's flag word and
        # is only worth its round trip when some status word carries the time-out pattern
        if iterations > 0 and bool(((out_st & poison) == poison).any().item()) \
                and split_error(dev.index):
            # some status words carry the time-out pattern and the handle's wait flag is up (the
            # groups of a big batch go out as several cooperative launches: one of them can time
            # out alone): a cooperative launch did not get its workgroups co-resident (other
            # kernels of this process held the compute units).  Not a numerical failure: say "not served"
            # and let the caller run the step-wise loop, which needs no co-residency.
            import warnings
            warnings.warn('cooperative shared-weight launch timed out waiting for co-residency; '
                          'repeating the fit step by step', RuntimeWarning, stacklevel=2)
            split_reset(dev.index)  # consume the report (see _checked_with_split_retry)
            return None
        _status_raise_em(out_st, 'CACGMMTrainer.fit')
    return dict(eigvec=out_vec, eigval=out_val, weight=out_w, status=out_st,
                affiliation=out_aff, quadratic_form=out_q)


def em_predict(y, eigvec, eigval, weight, *, activity=None,
               layout=_lib.LAYOUT_TD, affiliation_eps=0.0,
               want_q=False, want_log_pdf=False, want_affiliation=True):
    """pbbss_cacgmm_predict.  weight is (B,K) or (B,K,T) f64."""
    t = _t()
    dev = y.device
    if layout == _lib.LAYOUT_TD:
        B, T, D = y.shape
    else:
        B, D, T = y.shape
    K = eigvec.shape[1]
    is128 = y.dtype == t.complex128
    f64 = t.float64
    if weight.dim() == 2:
        wb, wk, wt = K, 1, 0
    else:
        assert weight.shape == (B, K, T), weight.shape
        wb, wk, wt = K * T, T, 1
    aff = t.empty((B, K, T), dtype=f64, device=dev) if want_affiliation else None
    q = t.empty((B, K, T), dtype=f64, device=dev) if want_q else None
    lp = t.empty((B, K, T), dtype=f64, device=dev) if want_log_pdf else None
    rc = _lib.load().pbbss_cacgmm_predict(
        _lib.handle(dev.index), _lib.ptr(y), B, T, D, K, _lib.ptr(eigvec),
        _lib.ptr(eigval), _lib.ptr(weight), wb, wk, wt, _lib.ptr(activity),
        int(layout), int(is128), float(affiliation_eps), _lib.ptr(aff),
        _lib.ptr(q), _lib.ptr(lp), _lib.stream_ptr(dev.index))
    _lib.check(rc, f'cacgmm_predict(B={B},T={T},D={D},K={K})')
    return aff, q, lp


def cacg_m_step(y, saliency, quadratic_form, *, layout=_lib.LAYOUT_DT,
                covariance_norm='eigenvalue', eigenvalue_floor=1e-10,
                want_cov=False, check_status=True):
    """pbbss_cacg_m_step: saliency/quadratic_form (B,K,T) f64."""
    t = _t()
    dev = y.device
    if layout == _lib.LAYOUT_TD:
        B, T, D = y.shape
    else:
        B, D, T = y.shape
    K = saliency.shape[1]
    is128 = y.dtype == t.complex128

    def launch():
        out_vec = t.empty((B, K, D, D), dtype=t.complex128, device=dev)
        out_val = t.empty((B, K, D), dtype=t.float64, device=dev)
        out_st = t.zeros((B, K), dtype=t.int32, device=dev)
        out_cov = t.empty((B, K, D, D), dtype=t.complex128, device=dev) if want_cov else None
        rc = _lib.load().pbbss_cacg_m_step(
            _lib.handle(dev.index), _lib.ptr(y), B, T, D, K, _lib.ptr(saliency),
            _lib.ptr(quadratic_form), int(layout), int(is128),
            _lib.COVNORM[covariance_norm], float(eigenvalue_floor),
            _lib.ptr(out_vec), _lib.ptr(out_val), _lib.ptr(out_cov),
            _lib.ptr(out_st), _lib.stream_ptr(dev.index))
        _lib.check(rc, f'cacg_m_step(B={B},T={T},D={D},K={K})')
        return dict(eigvec=out_vec, eigval=out_val, cov=out_cov, status=out_st)

    # ONE iteration of the EM kernel: split groups need kSplitMinIterations = 3 (em_launch.hpp), so
    # this launch has no inter-workgroup wait and no time-out to recover from
    r = launch()
    if check_status:
        _status_raise_em(r['status'], 'ComplexAngularCentralGaussianTrainer._fit')
    return r['eigvec'], r['eigval'], r['cov'], r['status']


def heev(a):
    """pbbss_heev_batched: a (N,D,D) c128 -> (eigval (N,D), eigvec (N,D,D), status)."""
    t = _t()
    N, D, _ = a.shape
    val = t.empty((N, D), dtype=t.float64, device=a.device)
    vec = t.empty((N, D, D), dtype=t.complex128, device=a.device)
    st = t.zeros((N,), dtype=t.int32, device=a.device)
    rc = _lib.load().pbbss_heev_batched(
        _lib.handle(a.device.index), _lib.ptr(a), N, D, _lib.ptr(val),
        _lib.ptr(vec), _lib.ptr(st), _lib.stream_ptr(a.device.index))
    _lib.check(rc, f'heev(N={N},D={D})')
    return val, vec, st


def psd(x, mask, normalize=True):
    """pbbss_psd: x (B,D,T) complex, mask (B,K,T) f64 or None -> (B,K,D,D) c128."""
    t = _t()
    B, D, T = x.shape
    K = 1 if mask is None else mask.shape[1]
    out = t.empty((B, K, D, D), dtype=t.complex128, device=x.device)
    rc = _lib.load().pbbss_psd(
        _lib.handle(x.device.index), _lib.ptr(x), int(x.dtype == t.complex128),
        B, T, D, K, _lib.ptr(mask), int(bool(normalize)), _lib.ptr(out),
        _lib.stream_ptr(x.device.index))
    _lib.check(rc, f'psd(B={B},T={T},D={D},K={K})')
    return out


def gev(target, noise):
    """pbbss_gev: (N,D,D) c128 x2 -> (w (N,D) c128, status (N,))."""
    t = _t()
    N, D, _ = target.shape
    w = t.empty((N, D), dtype=t.complex128, device=target.device)
    st = t.zeros((N,), dtype=t.int32, device=target.device)
    rc = _lib.load().pbbss_gev(
        _lib.handle(target.device.index), _lib.ptr(target), _lib.ptr(noise), N,
        D, _lib.ptr(w), _lib.ptr(st), _lib.stream_ptr(target.device.index))
    _lib.check(rc, f'gev(N={N},D={D})')
    return w, st


def gev_general(target, noise, want_eigenvalue=False):
    """pbbss_gev_general: (N,D,D) c128 x2, no Hermitian assumption -> (w (N,D) unit norm,
    eigenvalue (N,) c128 or None, status (N,))."""
    t = _t()
    N, D, _ = target.shape
    w = t.empty((N, D), dtype=t.complex128, device=target.device)
    lam = t.empty((N,), dtype=t.complex128, device=target.device) if want_eigenvalue else None
    st = t.zeros((N,), dtype=t.int32, device=target.device)
    rc = _lib.load().pbbss_gev_general(
        _lib.handle(target.device.index), _lib.ptr(target), _lib.ptr(noise), N, D, _lib.ptr(w),
        _lib.ptr(lam) if lam is not None else None, _lib.ptr(st),
        _lib.stream_ptr(target.device.index))
    _lib.check(rc, f'gev_general(N={N},D={D})')
    return w, lam, st


def solve(A, Bm):
    """pbbss_solve: A (N,D,D), Bm (N,D,M) c128 -> (X, status)."""
    t = _t()
    N, D, _ = A.shape
    M = Bm.shape[-1]
    x = t.empty((N, D, M), dtype=t.complex128, device=A.device)
    st = t.zeros((N,), dtype=t.int32, device=A.device)
    rc = _lib.load().pbbss_solve(
        _lib.handle(A.device.index), _lib.ptr(A), _lib.ptr(Bm), N, D, M,
        _lib.ptr(x), _lib.ptr(st), _lib.stream_ptr(A.device.index))
    _lib.check(rc, f'solve(N={N},D={D},M={M})')
    return x, st


def mvdr_souden(target, noise, eps):
    """pbbss_mvdr_souden -> (mat (N,D,D), snr_num (N,D), snr_den (N,D), status)."""
    t = _t()
    N, D, _ = target.shape
    mat = t.empty((N, D, D), dtype=t.complex128, device=target.device)
    num = t.empty((N, D), dtype=t.complex128, device=target.device)
    den = t.empty((N, D), dtype=t.complex128, device=target.device)
    # the fused kernels (D <= 8) write every status word; no fill launch in front of them
    st = (t.empty if D <= 8 else t.zeros)((N,), dtype=t.int32, device=target.device)
    rc = _lib.load().pbbss_mvdr_souden(
        _lib.handle(target.device.index), _lib.ptr(target), _lib.ptr(noise), N,
        D, float(eps), _lib.ptr(mat), _lib.ptr(num), _lib.ptr(den), _lib.ptr(st),
        _lib.stream_ptr(target.device.index))
    _lib.check(rc, f'mvdr_souden(N={N},D={D})')
    return mat, num, den, st


def select_reference_channel(mat, num, den, L, F, eps, lead_stride=None, bin_stride=1):
    """pbbss_select_reference_channel (round 6): the automatic reference channel of
    get_mvdr_vector_souden (beamformer.py:601-624) and the column select (:690-698) for L problems
    of F bins in ONE launch, from the outputs of `mvdr_souden` (mat (N,D,D), num / den (N,D) c128;
    problem l, bin f = matrix l * lead_stride + f * bin_stride).
    -> (w (L,F,D) c128, ref (L) int32, ok (L) int32: every SNR of the problem finite)."""
    t = _t()
    D = mat.shape[-1]
    w = t.empty((L, F, D), dtype=t.complex128, device=mat.device)
    ref = t.empty((L,), dtype=t.int32, device=mat.device)
    ok = t.empty((L,), dtype=t.int32, device=mat.device)
    rc = _lib.load().pbbss_select_reference_channel(
        _lib.handle(mat.device.index), _lib.ptr(mat), _lib.ptr(num), _lib.ptr(den), int(L), int(F),
        int(D), int(F if lead_stride is None else lead_stride), int(bin_stride), float(eps),
        _lib.ptr(w), _lib.ptr(ref), _lib.ptr(ok), _lib.stream_ptr(mat.device.index))
    _lib.check(rc, f'select_reference_channel(L={L},F={F},D={D})')
    return w, ref, ok


def mvdr(atf, noise):
    """pbbss_mvdr: atf (N,D), noise (N,D,D) c128 -> (w (N,D), status)."""
    t = _t()
    N, D = atf.shape
    w = t.empty((N, D), dtype=t.complex128, device=atf.device)
    st = t.zeros((N,), dtype=t.int32, device=atf.device)
    rc = _lib.load().pbbss_mvdr(
        _lib.handle(atf.device.index), _lib.ptr(atf), _lib.ptr(noise), N, D,
        _lib.ptr(w), _lib.ptr(st), _lib.stream_ptr(atf.device.index))
    _lib.check(rc, f'mvdr(N={N},D={D})')
    return w, st


def ban(w, noise):
    """pbbss_ban: w (N,D), noise (N,D,D) c128 -> (N,D) c128."""
    t = _t()
    N, D = w.shape
    out = t.empty((N, D), dtype=t.complex128, device=w.device)
    rc = _lib.load().pbbss_ban(
        _lib.handle(w.device.index), _lib.ptr(w), _lib.ptr(noise), N, D,
        _lib.ptr(out), _lib.stream_ptr(w.device.index))
    _lib.check(rc, f'ban(N={N},D={D})')
    return out


def apply_bf(w, x):
    """pbbss_apply_beamforming_vector: w (B,D) c128, x (B,D,T) -> (B,T) c128.  With fewer
    observations than vectors (x (Bx,D,T), B a multiple of Bx) problem b reads x[b % Bx] --
    pbbss_apply_beamforming_vector_shared, no copies of x."""
    t = _t()
    Bx, D, T = x.shape
    B = w.shape[0]
    out = t.empty((B, T), dtype=t.complex128, device=x.device)
    if B == Bx:
        rc = _lib.load().pbbss_apply_beamforming_vector(
            _lib.handle(x.device.index), _lib.ptr(w), _lib.ptr(x),
            int(x.dtype == t.complex128), B, T, D, _lib.ptr(out),
            _lib.stream_ptr(x.device.index))
    else:
        rc = _lib.load().pbbss_apply_beamforming_vector_shared(
            _lib.handle(x.device.index), _lib.ptr(w), _lib.ptr(x),
            int(x.dtype == t.complex128), B, Bx, T, D, _lib.ptr(out),
            _lib.stream_ptr(x.device.index))
    _lib.check(rc, f'apply_beamforming_vector(B={B},Bx={Bx},T={T},D={D})')
    return out


def set_timing(enable, device_index=None):
    _lib.check(_lib.load().pbbss_set_timing(_lib.handle(device_index), int(enable)), 'set_timing')


def last_kernel_ms(device_index=None, lag=0):
    """Duration of the timed region `lag` launches ago (0 = the most recent: waits for it).  Read
    with lag = 2 inside a loop of launches, the queue keeps two launches behind the running one
    (pbbss_kernel_ms_lagged)."""
    ms = ctypes.c_float()
    _lib.check(_lib.load().pbbss_kernel_ms_lagged(_lib.handle(device_index), int(lag),
                                                  ctypes.byref(ms)), 'kernel_ms_lagged')
    return float(ms.value)


def dhtv_calculate_mapping(mask, plan, optimal=False, metric='cos'):
    """pbbss_dhtv_calculate_mapping: mask (U,K,F,T) f64, plan int32 (P,3) on the
    device -> (mapping int32 (U,K,F), aligned unit-norm features (U,K,F,T), status (U,))."""
    t = _t()
    U, K, F, T = mask.shape
    feat = t.empty_like(mask)
    mapping = t.empty((U, K, F), dtype=t.int32, device=mask.device)
    st = t.zeros((U,), dtype=t.int32, device=mask.device)
    rc = _lib.load().pbbss_dhtv_calculate_mapping(
        _lib.handle(mask.device.index), _lib.ptr(mask), U, K, F, T, _lib.ptr(plan),
        int(plan.shape[0]), int(bool(optimal)), PA_METRIC[metric], _lib.ptr(feat), _lib.ptr(mapping),
        _lib.ptr(st), _lib.stream_ptr(mask.device.index))
    _lib.check(rc, f'dhtv_calculate_mapping(U={U},K={K},F={F},T={T})')
    return mapping, feat, st


def apply_mapping(mask, mapping):
    """pbbss_apply_mapping: mask (U,K,F,T) f64, mapping int32 (U,K,F) -> (U,K,F,T)."""
    t = _t()
    U, K, F, T = mask.shape
    out = t.empty_like(mask)
    rc = _lib.load().pbbss_apply_mapping(
        _lib.handle(mask.device.index), _lib.ptr(mask), _lib.ptr(mapping), U, K, F, T,
        _lib.ptr(out), _lib.stream_ptr(mask.device.index))
    _lib.check(rc, f'apply_mapping(U={U},K={K},F={F},T={T})')
    return out


PA_METRIC = {'cos': 0, 'multiply': 1, 'euclidean': 2}


def _strides3(x):
    """element strides of (utterance, class, bin) of a (U,K,F,T) tensor with contiguous frames"""
    assert x.stride(-1) == 1 or x.shape[-1] == 1, x.stride()
    return (ctypes.c_int64 * 3)(x.stride(0), x.stride(1), x.stride(2))


def _view_ptr(x):
    """device pointer of a (possibly strided) view; the strides travel separately"""
    assert x.is_cuda, x.device
    return ctypes.c_void_p(x.data_ptr())


def pa_pairwise_mapping(mask, reference, metric, optimal, *, mapping=None, col0=0,
                        want_scores=False):
    """pbbss_pa_pairwise_mapping: mask / reference (U,K,F,T) f64 views (frames contiguous)
    -> (mapping int32 (U,K,map_F) with columns col0 .. col0+F-1 filled, scores or None,
    status (U,))."""
    t = _t()
    U, K, F, T = mask.shape
    assert tuple(reference.shape) == (U, K, F, T), (mask.shape, reference.shape)
    if mapping is None:
        mapping = t.empty((U, K, col0 + F), dtype=t.int32, device=mask.device)
    scores = t.empty((U, F, K, K), dtype=t.float64, device=mask.device) if want_scores else None
    st = t.zeros((U,), dtype=t.int32, device=mask.device)
    rc = _lib.load().pbbss_pa_pairwise_mapping(
        _lib.handle(mask.device.index), _view_ptr(mask), _view_ptr(reference), U, K, F, T,
        _strides3(mask), _strides3(reference), PA_METRIC[metric], int(bool(optimal)),
        _lib.ptr(scores) if want_scores else None, _lib.ptr(mapping), int(mapping.shape[-1]),
        int(col0), _lib.ptr(st), _lib.stream_ptr(mask.device.index))
    _lib.check(rc, f'pa_pairwise_mapping(U={U},K={K},F={F},T={T},{metric})')
    return mapping, scores, st


def pa_compose_mapping(mapping):
    """pbbss_pa_compose_mapping, in place on mapping int32 (U,K,F)."""
    U, K, F = mapping.shape
    rc = _lib.load().pbbss_pa_compose_mapping(
        _lib.handle(mapping.device.index), _lib.ptr(mapping), U, K, F,
        _lib.stream_ptr(mapping.device.index))
    _lib.check(rc, f'pa_compose_mapping(U={U},K={K},F={F})')
    return mapping


def pa_mapping_from_scores(scores, optimal):
    """pbbss_pa_mapping_from_scores: scores (N,K,K) f64 -> (mapping int32 (K,N), status (1,))."""
    t = _t()
    N, K, _ = scores.shape
    mapping = t.empty((K, N), dtype=t.int32, device=scores.device)
    st = t.zeros((1,), dtype=t.int32, device=scores.device)
    rc = _lib.load().pbbss_pa_mapping_from_scores(
        _lib.handle(scores.device.index), _lib.ptr(scores), N, K, int(bool(optimal)),
        _lib.ptr(mapping), _lib.ptr(st), _lib.stream_ptr(scores.device.index))
    _lib.check(rc, f'pa_mapping_from_scores(N={N},K={K})')
    return mapping, st


def cwmm_fit(y, K, spline, *, gamma0=None, model=None, iterations=100, saliency=None,
             weight_mode=0, final_predict=False, want_log_pdf=False, check_status=True,
             group=None):
    """pbbss_cwmm_fit.  y (B,T,D) complex; gamma0 (B,K,T) f64 or
    model=(mode (B,K,D) c128, concentration (B,K), weight (B,K)).
    `spline` = dict(t, c device f64 arrays, ev_min, ev_max, max_concentration);
    may be None for a pure predict (iterations=0).
    group (with weight_mode=_lib.WEIGHT_SHARED_K): consecutive problems that share one weight
    set (weight_constant_axis (-3, -1)); the weight comes back as (B / group, K).  Returns None
    when the cooperative kernel does not serve the configuration or its grid barrier timed out
    (a RuntimeWarning): the caller then runs the loop step by step."""
    t = _t()
    dev = y.device
    B, T, D = y.shape
    is128 = y.dtype == t.complex128
    shared = weight_mode in (_lib.WEIGHT_SHARED_K, _lib.WEIGHT_SHARED_KT)
    opts = _lib.CwmmOpts(
        iterations=int(iterations), weight_mode=int(weight_mode), y_is_c128=int(is128),
        final_predict=int(bool(final_predict or want_log_pdf)),
        n_coef=0 if spline is None else int(spline['c'].numel()),
        group=int(group) if shared else 0,
        ev_min=0.0 if spline is None else float(spline['ev_min']),
        ev_max=0.0 if spline is None else float(spline['ev_max']),
        max_concentration=0.0 if spline is None else float(spline['max_concentration']))
    f64 = t.float64
    in_mode = in_conc = in_w = None
    if model is not None:
        in_mode, in_conc, in_w = model
        assert in_mode.shape == (B, K, D) and in_conc.shape == (B, K) and in_w.shape == (B, K)
    else:
        assert gamma0.shape == (B, K, T) and gamma0.dtype == f64

    def launch():
        out_mode = t.empty((B, K, D), dtype=t.complex128, device=dev)
        out_conc = t.empty((B, K), dtype=f64, device=dev)
        out_w = t.empty(((B // group, K, T) if weight_mode == _lib.WEIGHT_SHARED_KT else
                         (B // group, K)) if shared else (B, K), dtype=f64, device=dev)
        out_st = t.zeros((B, K), dtype=t.int32, device=dev)
        out_aff = t.empty((B, K, T), dtype=f64, device=dev) if final_predict else None
        out_lp = t.empty((B, K, T), dtype=f64, device=dev) if want_log_pdf else None
        rc = _lib.load().pbbss_cwmm_fit(
            _lib.handle(dev.index), _lib.ptr(y), B, T, D, K, _lib.ptr(gamma0), _lib.ptr(in_mode),
            _lib.ptr(in_conc), _lib.ptr(in_w), _lib.ptr(saliency), ctypes.byref(opts),
            None if spline is None else _lib.ptr(spline['t']),
            None if spline is None else _lib.ptr(spline['c']),
            _lib.ptr(out_mode), _lib.ptr(out_conc), _lib.ptr(out_w), _lib.ptr(out_st),
            _lib.ptr(out_aff), _lib.ptr(out_lp), _lib.stream_ptr(dev.index))
        if shared and rc == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(rc, f'cwmm_fit(B={B},T={T},D={D},K={K})')
        return dict(mode=out_mode, concentration=out_conc, weight=out_w, status=out_st,
                    affiliation=out_aff, log_pdf=out_lp)

    if shared:
        r = launch()
        if r is None:
            return None
        if check_status:
            poison = _lib.ST_NONFINITE | _lib.ST_EIG_NOCONV
            # .any(): the groups of a big batch go out as several cooperative launches, one of
            # which can time out alone
            if bool(((r['status'] & poison) == poison).any().item()) and split_error(dev.index):
                import warnings
                warnings.warn('cooperative shared-weight launch timed out waiting for '
                              'co-residency; repeating the fit step by step', RuntimeWarning,
                              stacklevel=2)
                split_reset(dev.index)
                return None
            _status_raise_em(r['status'], 'CWMMTrainer.fit')
        return r
    if check_status and iterations > 0:
        return _checked_with_split_retry(launch, dev, 'CWMMTrainer.fit')
    return launch()


def wmwf(target, noise, distortion_weight=1.0, frequency_dependent=False):
    """pbbss_wmwf -> (filter matrix (N,D,D), snr_num (N,D), snr_den (N,D), status)."""
    t = _t()
    N, D, _ = target.shape
    mat = t.empty((N, D, D), dtype=t.complex128, device=target.device)
    num = t.empty((N, D), dtype=t.complex128, device=target.device)
    den = t.empty((N, D), dtype=t.complex128, device=target.device)
    st = t.zeros((N,), dtype=t.int32, device=target.device)
    rc = _lib.load().pbbss_wmwf(
        _lib.handle(target.device.index), _lib.ptr(target), _lib.ptr(noise), N, D,
        float(distortion_weight), int(bool(frequency_dependent)), _lib.ptr(mat),
        _lib.ptr(num), _lib.ptr(den), _lib.ptr(st), _lib.stream_ptr(target.device.index))
    _lib.check(rc, f'wmwf(N={N},D={D})')
    return mat, num, den, st


_DHTV_PROBE = {}  # (device index, host thread) -> last set_dhtv_probe flag word (default 0)
_SPLIT_TAIL = {}  # (device index, host thread) -> last set_split_tail value (default: on)


def set_split_tail(enable, device_index=None):
    """pbbss_set_split_tail: toggle the split-bin handling of remainder problems."""
    _lib.check(_lib.load().pbbss_set_split_tail(_lib.handle(device_index), int(bool(enable))),
               'set_split_tail')
    _SPLIT_TAIL[_handle_key(device_index)] = bool(enable)


def split_tail(device_index=None):
    """The split-tail setting in force on this thread's handle (handles are created with it on)."""
    return _SPLIT_TAIL.get(_handle_key(device_index), True)


def set_spin_limit(polls, device_index=None):
    """pbbss_set_spin_limit (test knob): polls before a bounded inter-workgroup wait gives up;
    0 = defaults.  A tiny value provokes real time-outs of the split / team protocols."""
    _lib.check(_lib.load().pbbss_set_spin_limit(_lib.handle(device_index), int(polls)),
               'set_spin_limit')


def split_reset(device_index=None):
    """pbbss_split_reset: consume a reported time-out (re-zero the protocol counters and the
    sticky flag of split_error) after the device has drained."""
    _lib.check(_lib.load().pbbss_split_reset(_lib.handle(device_index)), 'split_reset')


_DHTV_TEAM = {}  # (device index, host thread) -- one C handle each -- -> last set_dhtv_team value


def _handle_key(device_index):
    import threading
    if device_index is None:
        device_index = _lib.require_gpu().cuda.current_device()
    return (device_index, threading.get_ident())


def set_dhtv_team(workgroups_per_utterance, device_index=None):
    """pbbss_set_dhtv_team: 0 automatic, 1 one workgroup per utterance, >= 2 frame-slice kernel,
    -2..-32 bin-chunk team kernel of that size."""
    _lib.check(_lib.load().pbbss_set_dhtv_team(_lib.handle(device_index),
                                               int(workgroups_per_utterance)), 'set_dhtv_team')
    _DHTV_TEAM[_handle_key(device_index)] = int(workgroups_per_utterance)


def set_dhtv_probe(enable, device_index=None):
    """pbbss_set_dhtv_probe: evaluate all plan segments at once in front of the DHTV plan and
    skip the plan when the masks are aligned already (same results; for callers that align
    every EM iteration).  `enable` True / False, or the flag word of the C call (2: the aligned
    features are not written, 3: probe + no features)."""
    flags = int(enable) if isinstance(enable, int) and not isinstance(enable, bool) else int(bool(enable))
    _lib.check(_lib.load().pbbss_set_dhtv_probe(_lib.handle(device_index), flags),
               'set_dhtv_probe')
    _DHTV_PROBE[_handle_key(device_index)] = flags


def dhtv_probe(device_index=None):
    """The probe flag word in force on this thread's handle (its last set_dhtv_probe, else 0)."""
    return _DHTV_PROBE.get(_handle_key(device_index), 0)


def dhtv_team(device_index=None):
    """The team setting in force on this thread's handle: its last set_dhtv_team, else
    PBBSS_DHTV_TEAM (read by the library when the handle is created), else 0 = automatic."""
    key = _handle_key(device_index)
    if key in _DHTV_TEAM:
        return _DHTV_TEAM[key]
    try:
        return int(os.environ.get('PBBSS_DHTV_TEAM', '0'))
    except ValueError:
        return 0


def split_error(device_index=None):
    """pbbss_split_error: 1 if an inter-workgroup wait of a split launch ever timed out."""
    flag = ctypes.c_int()
    _lib.check(_lib.load().pbbss_split_error(_lib.handle(device_index), ctypes.byref(flag)),
               'split_error')
    return int(flag.value)


# ---- N2/N3: real-embedding mixtures and joint spatial+spectral models ---------
def _real_embedding(y):
    """float32 stays float32 on the device (the kernels widen to float64 on load)."""
    t = _t()
    if y.dtype not in (t.float32, t.float64):
        y = y.to(t.float64)
    return y.contiguous()


def embed_log_pdf(y, kind, mean, scale):
    """pbbss_embed_log_pdf.  y (B,N,E) real; mean (B,K,E); scale (B,K) -> (B,K,N) f64."""
    t = _t()
    y = _real_embedding(y)
    B, N, E = y.shape
    K = mean.shape[1]
    out = t.empty((B, K, N), dtype=t.float64, device=y.device)
    rc = _lib.load().pbbss_embed_log_pdf(
        _lib.handle(y.device.index), _lib.ptr(y), int(y.dtype == t.float64), B, N, E, K, int(kind),
        _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(out), _lib.stream_ptr(y.device.index))
    _lib.check(rc, f'embed_log_pdf(B={B},N={N},E={E},K={K})')
    return out


def embed_fit(y, kind, weights, *, normalize=False, min_concentration=1e-10,
              max_concentration=500.):
    """pbbss_embed_fit.  y (B,N,E) real; weights (B,K,N) f64 -> mean (B,K,E), scale (B,K)
    ((B,K,E) per-dimension variances for EMBED_GAUSS_DIAG)."""
    t = _t()
    y = _real_embedding(y)
    B, N, E = y.shape
    K = weights.shape[1]
    assert weights.shape == (B, K, N) and weights.dtype == t.float64
    mean = t.empty((B, K, E), dtype=t.float64, device=y.device)
    scale = t.empty((B, K, E) if kind == _lib.EMBED_GAUSS_DIAG else (B, K), dtype=t.float64,
                    device=y.device)
    rc = _lib.load().pbbss_embed_fit(
        _lib.handle(y.device.index), _lib.ptr(y), int(y.dtype == t.float64), B, N, E, K, int(kind),
        int(bool(normalize)), _lib.ptr(weights), float(min_concentration),
        float(max_concentration), _lib.ptr(mean), _lib.ptr(scale), _lib.stream_ptr(y.device.index))
    _lib.check(rc, f'embed_fit(B={B},N={N},E={E},K={K})')
    return mean, scale


def vmfmm_fit(y, K, *, gamma0=None, model=None, iterations=100, saliency=None, weight_mode=0,
              min_concentration=1e-10, max_concentration=500., final_predict=False,
              want_log_pdf=False):
    """pbbss_vmfmm_fit.  y (B,N,E) real; gamma0 (B,K,N) f64 or (iterations=0)
    model=(mean (B,K,E), concentration (B,K), weight (B,K))."""
    t = _t()
    y = _real_embedding(y)
    dev = y.device
    B, N, E = y.shape
    f64 = t.float64
    opts = _lib.MixOpts(iterations=int(iterations), kind=_lib.EMBED_VMF,
                        weight_mode=int(weight_mode), embedding_is_f64=int(y.dtype == f64),
                        final_predict=int(bool(final_predict or want_log_pdf)),
                        min_concentration=float(min_concentration),
                        max_concentration=float(max_concentration))
    mean = t.empty((B, K, E), dtype=f64, device=dev)
    conc = t.empty((B, K), dtype=f64, device=dev)
    weight = t.empty((B, K), dtype=f64, device=dev)
    aff = t.empty((B, K, N), dtype=f64, device=dev) if final_predict else None
    lp = t.empty((B, K, N), dtype=f64, device=dev) if want_log_pdf else None
    in_mean = in_conc = in_w = None
    if model is not None:
        in_mean, in_conc, in_w = model
        assert in_mean.shape == (B, K, E) and in_conc.shape == (B, K) and in_w.shape == (B, K)
    else:
        assert gamma0.shape == (B, K, N) and gamma0.dtype == f64
    rc = _lib.load().pbbss_vmfmm_fit(
        _lib.handle(dev.index), _lib.ptr(y), B, N, E, K, _lib.ptr(gamma0), _lib.ptr(in_mean),
        _lib.ptr(in_conc), _lib.ptr(in_w), _lib.ptr(saliency), ctypes.byref(opts),
        _lib.ptr(mean), _lib.ptr(conc), _lib.ptr(weight), _lib.ptr(aff), _lib.ptr(lp),
        _lib.stream_ptr(dev.index))
    _lib.check(rc, f'vmfmm_fit(B={B},N={N},E={E},K={K})')
    return dict(mean=mean, concentration=conc, weight=weight, affiliation=aff, log_pdf=lp)


def gauss_full_fit(y, weights):
    """pbbss_gauss_full_fit.  y (B,N,E) real; weights (B,K,N) f64 -> (mean (B,K,E), cov (B,K,E,E))."""
    t = _t()
    y = _real_embedding(y)
    B, N, E = y.shape
    K = weights.shape[1]
    assert weights.shape == (B, K, N) and weights.dtype == t.float64, (weights.shape, y.shape)
    mean = t.empty((B, K, E), dtype=t.float64, device=y.device)
    cov = t.empty((B, K, E, E), dtype=t.float64, device=y.device)
    rc = _lib.load().pbbss_gauss_full_fit(
        _lib.handle(y.device.index), _lib.ptr(y), int(y.dtype == t.float64), B, N, E, K,
        _lib.ptr(weights), _lib.ptr(mean), _lib.ptr(cov), _lib.stream_ptr(y.device.index))
    _lib.check(rc, f'gauss_full_fit(B={B},N={N},E={E},K={K})')
    return mean, cov


def gauss_full_log_pdf(y, mean, cov):
    """pbbss_gauss_full_log_pdf.  y (B,N,E); mean (B,K,E); cov (B,K,E,E) -> ((B,K,N), status)."""
    t = _t()
    y = _real_embedding(y)
    B, N, E = y.shape
    K = mean.shape[1]
    assert mean.shape == (B, K, E) and cov.shape == (B, K, E, E), (mean.shape, cov.shape)
    out = t.empty((B, K, N), dtype=t.float64, device=y.device)
    st = t.zeros((1,), dtype=t.int32, device=y.device)
    rc = _lib.load().pbbss_gauss_full_log_pdf(
        _lib.handle(y.device.index), _lib.ptr(y), int(y.dtype == t.float64), B, N, E, K,
        _lib.ptr(mean), _lib.ptr(cov), _lib.ptr(out), _lib.ptr(st),
        _lib.stream_ptr(y.device.index))
    _lib.check(rc, f'gauss_full_log_pdf(B={B},N={N},E={E},K={K})')
    return out, st


def gmm_fit(y, K, *, gamma0=None, model=None, iterations=100, saliency=None, weight_mode=0,
            fixed_covariance=None, final_predict=False, want_log_pdf=False):
    """pbbss_gmm_fit (spherical covariances).  y (B,N,E) real, used as given (no row
    normalisation); gamma0 (B,K,N) f64 or (iterations=0) model=(mean (B,K,E),
    covariance (B,K), weight (B,K)); fixed_covariance (B,K) or None."""
    t = _t()
    y = _real_embedding(y)
    dev = y.device
    B, N, E = y.shape
    f64 = t.float64
    opts = _lib.MixOpts(iterations=int(iterations), kind=_lib.EMBED_GAUSS_SPHERICAL,
                        weight_mode=int(weight_mode), embedding_is_f64=int(y.dtype == f64),
                        final_predict=int(bool(final_predict or want_log_pdf)))
    mean = t.empty((B, K, E), dtype=f64, device=dev)
    cov = t.empty((B, K), dtype=f64, device=dev)
    weight = t.empty((B, K), dtype=f64, device=dev)
    aff = t.empty((B, K, N), dtype=f64, device=dev) if final_predict else None
    lp = t.empty((B, K, N), dtype=f64, device=dev) if want_log_pdf else None
    in_mean = in_cov = in_w = None
    if model is not None:
        in_mean, in_cov, in_w = model
        assert in_mean.shape == (B, K, E) and in_cov.shape == (B, K) and in_w.shape == (B, K)
    else:
        assert gamma0.shape == (B, K, N) and gamma0.dtype == f64
    if fixed_covariance is not None:
        assert fixed_covariance.shape == (B, K) and fixed_covariance.dtype == f64
    rc = _lib.load().pbbss_gmm_fit(
        _lib.handle(dev.index), _lib.ptr(y), B, N, E, K, _lib.ptr(gamma0), _lib.ptr(in_mean),
        _lib.ptr(in_cov), _lib.ptr(in_w), _lib.ptr(saliency), _lib.ptr(fixed_covariance),
        ctypes.byref(opts), _lib.ptr(mean), _lib.ptr(cov), _lib.ptr(weight), _lib.ptr(aff),
        _lib.ptr(lp), _lib.stream_ptr(dev.index))
    _lib.check(rc, f'gmm_fit(B={B},N={N},E={E},K={K})')
    return dict(mean=mean, covariance=cov, weight=weight, affiliation=aff, log_pdf=lp)


def gmm_full_fit(y, K, *, gamma0=None, model=None, iterations=100, saliency=None, weight_mode=0,
                 fixed_covariance=None, final_predict=False, want_log_pdf=False):
    """pbbss_gmm_full_fit (full covariances, E <= 63).  As gmm_fit with covariance (B,K,E,E).
    Returns the status word too (non-zero: a covariance was not positive definite)."""
    t = _t()
    y = _real_embedding(y)
    dev = y.device
    B, N, E = y.shape
    f64 = t.float64
    opts = _lib.MixOpts(iterations=int(iterations), kind=_lib.EMBED_GAUSS_SPHERICAL,
                        weight_mode=int(weight_mode), embedding_is_f64=int(y.dtype == f64),
                        final_predict=int(bool(final_predict or want_log_pdf)))
    mean = t.empty((B, K, E), dtype=f64, device=dev)
    cov = t.empty((B, K, E, E), dtype=f64, device=dev)
    weight = t.empty((B, K), dtype=f64, device=dev)
    aff = t.empty((B, K, N), dtype=f64, device=dev) if final_predict else None
    lp = t.empty((B, K, N), dtype=f64, device=dev) if want_log_pdf else None
    st = t.zeros((1,), dtype=t.int32, device=dev)
    in_mean = in_cov = in_w = None
    if model is not None:
        in_mean, in_cov, in_w = model
        assert in_mean.shape == (B, K, E) and in_cov.shape == (B, K, E, E) and in_w.shape == (B, K)
    else:
        assert gamma0.shape == (B, K, N) and gamma0.dtype == f64
    if fixed_covariance is not None:
        assert fixed_covariance.shape == (B, K, E, E) and fixed_covariance.dtype == f64
    rc = _lib.load().pbbss_gmm_full_fit(
        _lib.handle(dev.index), _lib.ptr(y), B, N, E, K, _lib.ptr(gamma0), _lib.ptr(in_mean),
        _lib.ptr(in_cov), _lib.ptr(in_w), _lib.ptr(saliency), _lib.ptr(fixed_covariance),
        ctypes.byref(opts), _lib.ptr(mean), _lib.ptr(cov), _lib.ptr(weight), _lib.ptr(aff),
        _lib.ptr(lp), _lib.ptr(st), _lib.stream_ptr(dev.index))
    _lib.check(rc, f'gmm_full_fit(B={B},N={N},E={E},K={K})')
    return dict(mean=mean, covariance=cov, weight=weight, affiliation=aff, log_pdf=lp, status=st)


def estimate_mixture_weight(affiliation, saliency, reduce_inner, reduce_n):
    """pbbss_estimate_mixture_weight: affiliation (Bo, Bi, K, N) f64, saliency (Bo, Bi, N) or
    None -> (Bo, 1 if reduce_inner else Bi, K, 1 if reduce_n else N); None where the kernel does
    not serve the shape (a saliency with more than 16 classes): the caller takes the host formula."""
    t = _t()
    Bo, Bi, K, N = affiliation.shape
    out = t.empty((Bo, 1 if reduce_inner else Bi, K, 1 if reduce_n else N), dtype=t.float64,
                  device=affiliation.device)
    rc = _lib.load().pbbss_estimate_mixture_weight(
        _lib.handle(affiliation.device.index), _lib.ptr(affiliation), _lib.ptr(saliency), Bo, Bi, K,
        N, int(bool(reduce_inner)), int(bool(reduce_n)), _lib.ptr(out),
        _lib.stream_ptr(affiliation.device.index))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, f'estimate_mixture_weight(Bo={Bo},Bi={Bi},K={K},N={N})')
    return out


def log_pdf_to_affiliation(log_pdf, weight, activity=None, affiliation_eps=0.):
    """pbbss_log_pdf_to_affiliation: log_pdf (B,K,N) f64; weight a tensor that broadcasts against
    it -- (B,K,1), (K,1), (1,K,N), (B,1,N) ...: singleton axes become zero strides -> (B,K,N)."""
    t = _t()
    B, K, N = log_pdf.shape
    w = weight.to(t.float64)
    while w.ndim < 3:
        w = w.unsqueeze(0)
    assert w.ndim == 3 and all(a in (1, b) for a, b in zip(w.shape, (B, K, N))), (w.shape, (B, K, N))
    w = w.contiguous()
    st = [0 if w.shape[i] == 1 else w.stride(i) for i in range(3)]
    out = t.empty((B, K, N), dtype=t.float64, device=log_pdf.device)
    if activity is not None:
        assert activity.shape == (B, K, N) and activity.dtype == t.uint8
    rc = _lib.load().pbbss_log_pdf_to_affiliation(
        _lib.handle(log_pdf.device.index), _lib.ptr(log_pdf.contiguous()), B, K, N, _lib.ptr(w),
        st[0], st[1], st[2], _lib.ptr(activity), float(affiliation_eps), _lib.ptr(out),
        _lib.stream_ptr(log_pdf.device.index))
    _lib.check(rc, f'log_pdf_to_affiliation(B={B},K={K},N={N})')
    return out


def _broadcast_weight(weight, shape):
    """weight -> (contiguous float64 tensor, strides with 0 on singleton axes) against `shape`."""
    t = _t()
    w = weight.to(t.float64)
    while w.ndim < len(shape):
        w = w.unsqueeze(0)
    assert w.ndim == len(shape) and all(a in (1, b) for a, b in zip(w.shape, shape)), \
        (tuple(w.shape), tuple(shape))
    w = w.contiguous()
    return w, [0 if w.shape[i] == 1 else w.stride(i) for i in range(w.ndim)]


def log_pdf_to_affiliation_inline_pa(spatial_log_pdf, spectral_log_pdf, weight, activity=None,
                                     affiliation_eps=0., want_permutation=False):
    """pbbss_log_pdf_to_affiliation_inline_pa: both log-pdfs (F,K,T) f64, weight broadcastable."""
    t = _t()
    F, K, T = spatial_log_pdf.shape
    assert spectral_log_pdf.shape == (F, K, T), (spectral_log_pdf.shape, (F, K, T))
    if K > 6:
        raise NotImplementedError(f'inline permutation alignment searches K! permutations per bin: '
                                  f'K <= 6 is served, got K={K}')
    w, st = _broadcast_weight(weight, (F, K, T))
    dev = spatial_log_pdf.device
    out = t.empty((F, K, T), dtype=t.float64, device=dev)
    perm = t.empty((F, K), dtype=t.int32, device=dev) if want_permutation else None
    if activity is not None:
        assert activity.shape == (F, K, T) and activity.dtype == t.uint8
    rc = _lib.load().pbbss_log_pdf_to_affiliation_inline_pa(
        _lib.handle(dev.index), _lib.ptr(spatial_log_pdf.to(t.float64).contiguous()),
        _lib.ptr(spectral_log_pdf.to(t.float64).contiguous()), F, K, T, _lib.ptr(w), st[0], st[1],
        st[2], _lib.ptr(activity), float(affiliation_eps), _lib.ptr(out), _lib.ptr(perm),
        _lib.stream_ptr(dev.index))
    _lib.check(rc, f'log_pdf_to_affiliation_inline_pa(F={F},K={K},T={T})')
    return (out, perm) if want_permutation else out


def joint_weight_shape(weight_mode, F, K, T):
    return {_lib.JOINT_WEIGHT_FK: (F, K), _lib.JOINT_WEIGHT_UNIFORM: (), _lib.JOINT_WEIGHT_K: (K,),
            _lib.JOINT_WEIGHT_KT: (K, T), _lib.JOINT_WEIGHT_CONST: ()}[weight_mode]


def joint_fit(observation, embedding, K, kind, *, gamma0=None, model=None, iterations=100,
              saliency=None, weight_mode=0, covariance_norm=1, eigenvalue_floor=1e-10,
              affiliation_eps=1e-10, spatial_weight=1., spectral_weight=1., inline_pa=False,
              min_concentration=1e-10, max_concentration=500., fixed_scale=None,
              final_predict=False, check_status=True, sharded=False):
    """pbbss_joint_fit.  observation (F,T,D) complex, embedding (F,T,E) real;
    gamma0 (F,K,T) f64 or (iterations=0) model=(eigvec, eigval, weight, mean (K,E), scale);
    scale: (K,) concentration / spherical variance, (K,E) diagonal variances, (K,E,E) full
    covariance, by `kind`."""
    t = _t()
    embedding = _real_embedding(embedding)
    dev = observation.device
    F, T, D = observation.shape
    E = embedding.shape[-1]
    assert embedding.shape == (F, T, E)
    f64 = t.float64
    opts = _lib.MixOpts(iterations=int(iterations), kind=int(kind), weight_mode=int(weight_mode),
                        embedding_is_f64=int(embedding.dtype == f64),
                        obs_is_c128=int(observation.dtype == t.complex128),
                        final_predict=int(bool(final_predict)), inline_pa=int(bool(inline_pa)),
                        covariance_norm=int(covariance_norm),
                        min_concentration=float(min_concentration),
                        max_concentration=float(max_concentration),
                        affiliation_eps=float(affiliation_eps),
                        eigenvalue_floor=float(eigenvalue_floor),
                        spatial_weight=float(spatial_weight),
                        spectral_weight=float(spectral_weight), sharded=int(bool(sharded)))
    wshape = joint_weight_shape(weight_mode, F, K, T)
    eigvec = t.empty((F, K, D, D), dtype=t.complex128, device=dev)
    eigval = t.empty((F, K, D), dtype=f64, device=dev)
    weight = t.empty(wshape, dtype=f64, device=dev)
    mean = t.empty((K, E), dtype=f64, device=dev)
    scale_shape = {_lib.EMBED_GAUSS_FULL: (K, E, E), _lib.EMBED_GAUSS_DIAG: (K, E)}.get(kind, (K,))
    scale = t.empty(scale_shape, dtype=f64, device=dev)
    status = t.zeros((F, K), dtype=t.int32, device=dev)
    aff = t.empty((F, K, T), dtype=f64, device=dev) if final_predict else None
    in_vec = in_val = in_w = in_mean = None
    in_scale = fixed_scale
    if model is not None:
        in_vec, in_val, in_w, in_mean, in_scale = model
        assert in_vec.shape == (F, K, D, D) and in_val.shape == (F, K, D)
        assert tuple(in_w.shape) == tuple(wshape) and in_mean.shape == (K, E)
        assert tuple(in_scale.shape) == scale_shape, (tuple(in_scale.shape), scale_shape)
    else:
        assert gamma0.shape == (F, K, T) and gamma0.dtype == f64
    rc = _lib.load().pbbss_joint_fit(
        _lib.handle(dev.index), _lib.ptr(observation), _lib.ptr(embedding), F, T, D, E, K,
        _lib.ptr(gamma0), _lib.ptr(in_vec), _lib.ptr(in_val), _lib.ptr(in_w), _lib.ptr(in_mean),
        _lib.ptr(in_scale), _lib.ptr(saliency), ctypes.byref(opts), _lib.ptr(eigvec),
        _lib.ptr(eigval), _lib.ptr(weight), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(status),
        _lib.ptr(aff), _lib.stream_ptr(dev.index))
    _lib.check(rc, f'joint_fit(F={F},T={T},D={D},E={E},K={K})')
    if kind == _lib.EMBED_GAUSS_FULL and int(status[0, 0].item()) & _lib.ST_NOT_POSDEF:
        raise ValueError(  # sklearn's _compute_precision_cholesky via gaussian.py:26
            'Fitting the mixture model failed because some components have ill-defined empirical '
            'covariance (not positive definite)')
    if check_status and iterations > 0:
        _status_raise_em(status, 'joint model fit')
    return dict(eigvec=eigvec, eigval=eigval, weight=weight, mean=mean, scale=scale,
                status=status, affiliation=aff)


# ---- N4: remaining beamformer family ------------------------------------------
def lcmv(atf, response, noise):
    """pbbss_lcmv.  atf (K,F,D), response (K), noise (F,D,D) c128 -> w (F,D), status (F)."""
    t = _t()
    K, F, D = atf.shape
    w = t.empty((F, D), dtype=t.complex128, device=atf.device)
    st = t.zeros((F,), dtype=t.int32, device=atf.device)
    rc = _lib.load().pbbss_lcmv(_lib.handle(atf.device.index), _lib.ptr(atf), _lib.ptr(response),
                                _lib.ptr(noise), F, D, K, _lib.ptr(w), _lib.ptr(st),
                                _lib.stream_ptr(atf.device.index))
    _lib.check(rc, f'lcmv(K={K},F={F},D={D})')
    return w, st


def phase_correction(vector):
    """pbbss_phase_correction.  vector (..., F, D) c128 contiguous."""
    t = _t()
    F, D = vector.shape[-2:]
    two_d = vector.ndim == 2
    lead = 1 if two_d else vector.shape[0]
    rest = 1 if two_d else int(np.prod(vector.shape[1:-2], dtype=np.int64))
    out = t.empty_like(vector)
    scratch = t.empty((max(lead * rest * (F - 1), 1),), dtype=t.complex128, device=vector.device)
    rc = _lib.load().pbbss_phase_correction(
        _lib.handle(vector.device.index), _lib.ptr(vector), lead, rest, F, D, int(two_d),
        _lib.ptr(scratch), _lib.ptr(out), _lib.stream_ptr(vector.device.index))
    _lib.check(rc, f'phase_correction(shape={tuple(vector.shape)})')
    return out


def snr_postfilter(w, target, noise):
    t = _t()
    F, D = w.shape
    out = t.empty((F,), dtype=t.complex128, device=w.device)
    rc = _lib.load().pbbss_snr_postfilter(_lib.handle(w.device.index), _lib.ptr(w),
                                          _lib.ptr(target), _lib.ptr(noise), F, D, _lib.ptr(out),
                                          _lib.stream_ptr(w.device.index))
    _lib.check(rc, f'snr_postfilter(F={F},D={D})')
    return out


def reference_channel_terms(w_mat, target, noise):
    """pbbss_reference_channel_terms.  (F,D,D) c128 x 3 -> num, den (F,D) c128."""
    t = _t()
    F, D, _ = w_mat.shape
    num = t.empty((F, D), dtype=t.complex128, device=w_mat.device)
    den = t.empty((F, D), dtype=t.complex128, device=w_mat.device)
    rc = _lib.load().pbbss_reference_channel_terms(
        _lib.handle(w_mat.device.index), _lib.ptr(w_mat), _lib.ptr(target), _lib.ptr(noise), F, D,
        _lib.ptr(num), _lib.ptr(den), _lib.stream_ptr(w_mat.device.index))
    _lib.check(rc, f'reference_channel_terms(F={F},D={D})')
    return num, den


def rank_one_approximation(covariance, vector):
    """pbbss_rank_one_approximation.  (N,D,D), (N,D) c128 -> (N,D,D)."""
    t = _t()
    N, D = vector.shape
    out = t.empty((N, D, D), dtype=t.complex128, device=vector.device)
    rc = _lib.load().pbbss_rank_one_approximation(
        _lib.handle(vector.device.index), _lib.ptr(covariance), _lib.ptr(vector), N, D,
        _lib.ptr(out), _lib.stream_ptr(vector.device.index))
    _lib.check(rc, f'rank_one_approximation(N={N},D={D})')
    return out


def matvec(matrix, vector):
    """pbbss_matvec.  (N,D,D), (N,D) c128 -> (N,D)."""
    t = _t()
    N, D = vector.shape
    out = t.empty((N, D), dtype=t.complex128, device=vector.device)
    rc = _lib.load().pbbss_matvec(_lib.handle(vector.device.index), _lib.ptr(matrix),
                                  _lib.ptr(vector), N, D, _lib.ptr(out),
                                  _lib.stream_ptr(vector.device.index))
    _lib.check(rc, f'matvec(N={N},D={D})')
    return out


def distortionless_normalization(w, atf, noise):
    t = _t()
    F, D = w.shape
    out = t.empty((F, D), dtype=t.complex128, device=w.device)
    rc = _lib.load().pbbss_distortionless_normalization(
        _lib.handle(w.device.index), _lib.ptr(w), _lib.ptr(atf), _lib.ptr(noise), F, D,
        _lib.ptr(out), _lib.stream_ptr(w.device.index))
    _lib.check(rc, f'distortionless_normalization(F={F},D={D})')
    return out


def zero_degree_normalization(vector, reference_channel):
    t = _t()
    N, D = vector.shape
    out = t.empty_like(vector)
    rc = _lib.load().pbbss_zero_degree_normalization(
        _lib.handle(vector.device.index), _lib.ptr(vector), N, D, int(reference_channel) % D,
        _lib.ptr(out), _lib.stream_ptr(vector.device.index))
    _lib.check(rc, f'zero_degree_normalization(N={N},D={D})')
    return out


def condition_covariance(x, gamma):
    t = _t()
    N, D, _ = x.shape
    out = t.empty_like(x)
    rc = _lib.load().pbbss_condition_covariance(_lib.handle(x.device.index), _lib.ptr(x), N, D,
                                                float(gamma), _lib.ptr(out),
                                                _lib.stream_ptr(x.device.index))
    _lib.check(rc, f'condition_covariance(N={N},D={D})')
    return out


def apply_online_bf(vector, mix):
    """pbbss_apply_online_beamforming_vector.  vector (T,F,D) c128, mix (F,D,T) c64/c128."""
    t = _t()
    T, F, D = vector.shape
    out = t.empty((F, T), dtype=t.complex128, device=mix.device)
    rc = _lib.load().pbbss_apply_online_beamforming_vector(
        _lib.handle(mix.device.index), _lib.ptr(vector), _lib.ptr(mix),
        int(mix.dtype == t.complex128), F, T, D, _lib.ptr(out), _lib.stream_ptr(mix.device.index))
    _lib.check(rc, f'apply_online_bf(T={T},F={F},D={D})')
    return out


def stft_num_frames(num_samples, size, shift, window_length, fading=True, pad=True):
    """pbbss_stft_num_frames (host arithmetic only)."""
    n = _lib.load().pbbss_stft_num_frames(int(num_samples), int(size), int(shift),
                                          int(window_length), int(bool(fading)), int(bool(pad)))
    if n < 0:
        _lib.check(n, f'stft_num_frames(N={num_samples},size={size},shift={shift})')
    return n


def stft(x, size, shift, window, *, fading=True, pad=True, layout=0, out_c128=True):
    """pbbss_stft.  x (C,N) float32/float64, window (window_length) f64 on the device.

    Returns (C,T,F) [layout 0] or (F,T,C) [layout 1] complex128 / complex64, F = size//2+1."""
    t = _t()
    C, N = x.shape
    wl = int(window.shape[0])
    T = stft_num_frames(N, size, shift, wl, fading, pad)
    if T <= 0:
        raise ValueError(f'stft: no complete frame in {N} samples (size={size}, pad={pad})')
    F = size // 2 + 1
    shape = (C, T, F) if layout == 0 else (F, T, C)
    out = t.empty(shape, dtype=t.complex128 if out_c128 else t.complex64, device=x.device)
    rc = _lib.load().pbbss_stft(
        _lib.handle(x.device.index), _lib.ptr(x), int(x.dtype == t.float64), C, N, int(size),
        int(shift), wl, _lib.ptr(window), int(bool(fading)), int(bool(pad)), int(layout),
        int(bool(out_c128)), _lib.ptr(out), _lib.stream_ptr(x.device.index))
    _lib.check(rc, f'stft(C={C},N={N},size={size},shift={shift},window_length={wl})')
    return out


def istft(X, size, shift, synthesis_window, *, fading=True):
    """pbbss_istft.  X (C,T,F) complex64/complex128 -> (C, n_out) float64."""
    t = _t()
    C, T, F = X.shape
    wl = int(synthesis_window.shape[0])
    n_out = T * shift + wl - shift - (2 * (wl - shift) if fading else 0)
    if n_out <= 0:
        raise ValueError(f'istft: {T} frames give no output samples')
    out = t.empty((C, n_out), dtype=t.float64, device=X.device)
    rc = _lib.load().pbbss_istft(
        _lib.handle(X.device.index), _lib.ptr(X), int(X.dtype == t.complex128), C, T, int(size),
        int(shift), wl, _lib.ptr(synthesis_window), int(bool(fading)), _lib.ptr(out), n_out,
        _lib.stream_ptr(X.device.index))
    _lib.check(rc, f'istft(C={C},T={T},size={size},shift={shift},window_length={wl})')
    return out
