"""GPU parity of the small batched solvers and streaming kernels against
NumPy/SciPy on the same seeded inputs (through the C ABI via pb_bss_amd.engine).
Tolerances: float64 kernels vs LAPACK, stated per test."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


def _dev(x):
    from pb_bss_amd import _lib
    return _lib.to_device(x)


def _host(x):
    from pb_bss_amd import _lib
    return _lib.to_host(x)


def _rand_c(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def _posdef(rng, N, D):
    X = _rand_c(rng, N, D, D + 2)
    return X @ X.conj().swapaxes(-1, -2) / (D + 2)


@pytest.mark.parametrize('D', [2, 3, 4, 5, 6, 7, 8])
def test_heev_matches_eigh(D):
    from pb_bss_amd import engine
    rng = np.random.default_rng(D)
    N = 257
    A = _posdef(rng, N, D)
    A[1] = np.eye(D)                      # degenerate spectrum
    A[2] = np.diag(np.arange(1, D + 1))   # already diagonal
    A[3, :, :] = np.outer(np.ones(D), np.ones(D))  # rank one
    val, vec, st = engine.heev(_dev(A))
    val, vec, st = _host(val), _host(vec), _host(st)
    assert (st == 0).all()
    ref = np.linalg.eigvalsh(A)
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    assert np.abs(val - ref).max() / scale.max() < 1e-13
    assert (np.diff(val, axis=-1) >= 0).all(), 'ascending like numpy.linalg.eigh'
    rec = np.einsum('nij,nj,nkj->nik', vec, val, vec.conj())
    assert np.abs(rec - A).max() < 1e-12 * max(1.0, np.abs(A).max())
    eye = np.einsum('nji,njk->nik', vec.conj(), vec)
    assert np.abs(eye - np.eye(D)).max() < 1e-13


@pytest.mark.parametrize('D,M', [(2, 2), (3, 1), (6, 6), (8, 8), (8, 1)])
def test_solve_matches_numpy(D, M):
    from pb_bss_amd import engine, _lib
    rng = np.random.default_rng(10 * D + M)
    N = 130
    A = _rand_c(rng, N, D, D)
    B = _rand_c(rng, N, D, M)
    A[5, 0, :] = 0                        # exactly singular
    x, st = engine.solve(_dev(A), _dev(B))
    x, st = _host(x), _host(st)
    assert st[5] == _lib.ST_SINGULAR
    ok = np.ones(N, bool)
    ok[5] = False
    assert (st[ok] == 0).all()
    ref = np.linalg.solve(A[ok], B[ok])
    assert np.abs(x[ok] - ref).max() / np.abs(ref).max() < 1e-10


@pytest.mark.parametrize('D', [2, 3, 6, 8])
def test_gev_matches_scipy(D):
    """cosine similarity as in the reference's own test
    (tests/test_extraction/test_beamformer.py:121-147, atol 1e-6)."""
    from pb_bss_amd import engine, _lib
    rng = np.random.default_rng(100 + D)
    N = 513
    T_ = _posdef(rng, N, D)
    Nn = _posdef(rng, N, D) + 0.1 * np.eye(D)
    Nn[7] = -np.eye(D)                    # not positive definite -> status
    w, st = engine.gev(_dev(T_), _dev(Nn))
    w, st = _host(w), _host(st)
    assert st[7] & _lib.ST_NOT_POSDEF and (st[7] >> 8) == 1
    ok = np.ones(N, bool)
    ok[7] = False
    assert (st[ok] == 0).all()
    for f in np.flatnonzero(ok)[::16]:
        vals, vecs = scipy.linalg.eigh(T_[f], Nn[f])
        r = vecs[:, -1]
        cos = abs(np.vdot(r, w[f])) / np.linalg.norm(r) / np.linalg.norm(w[f])
        assert abs(cos - 1) < 1e-9
        assert abs(np.vdot(w[f], Nn[f] @ w[f]).real - 1) < 1e-9  # zhegvd normalisation
    # GEV == PCA when the noise PSD is the identity (test_beamformer.py:98-104)
    I = np.broadcast_to(np.eye(D, dtype=complex), T_.shape).copy()
    w2, st2 = engine.gev(_dev(T_), _dev(I))
    w2 = _host(w2)
    for f in range(0, N, 37):
        v = np.linalg.eigh(T_[f])[1][:, -1]
        assert abs(abs(np.vdot(v, w2[f])) - 1) < 1e-9


@pytest.mark.parametrize('dtype', [np.complex64, np.complex128])
def test_normalize_observation(dtype):
    from pb_bss_amd import engine
    from oracle import cacgmm as oc
    rng = np.random.default_rng(0)
    y = _rand_c(rng, 7, 131, 6).astype(dtype)
    y[2, 5] = 0                           # zero frame stays zero (utils.py:251)
    out = _host(engine.normalize_observation(_dev(y)))
    ref = oc.normalize_observation(y)
    assert out.shape == ref.shape and out.dtype == ref.dtype
    tol = 3e-7 if dtype == np.complex64 else 1e-15
    assert np.abs(out - ref).max() < tol
    assert (out[2, :, 5] == 0).all()


@pytest.mark.parametrize('D,K', [(2, 1), (3, 2), (6, 3), (8, 2), (8, 6)])
def test_psd_matches_oracle(D, K):
    from pb_bss_amd import engine
    from oracle import beamformer as ob
    rng = np.random.default_rng(D * 10 + K)
    B, T = 33, 211
    x = _rand_c(rng, B, D, T).astype(np.complex64)
    mask = rng.uniform(size=(B, K, T))
    out = _host(engine.psd(_dev(x), _dev(mask), normalize=True))
    ref = ob.psd(x.astype(np.complex128), mask)
    assert np.abs(out - ref).max() < 1e-13 * np.abs(ref).max() + 1e-15
    out = _host(engine.psd(_dev(x), _dev(mask), normalize=False))
    ref = ob.psd(x.astype(np.complex128), mask, normalize=False)
    assert np.abs(out - ref).max() < 1e-12 * np.abs(ref).max()
    out = _host(engine.psd(_dev(x), None))
    ref = ob.psd(x.astype(np.complex128))
    assert np.abs(out[:, 0] - ref).max() < 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize('D', [2, 6, 8])
def test_mvdr_ban_apply(D):
    from pb_bss_amd import engine
    from oracle import beamformer as ob
    rng = np.random.default_rng(200 + D)
    N, T = 129, 97
    Pt = _posdef(rng, N, D)
    Pn = _posdef(rng, N, D) + 0.05 * np.eye(D)
    h = _rand_c(rng, N, D)
    w, st = engine.mvdr(_dev(h), _dev(Pn))
    ref = ob.mvdr(h, Pn)
    assert (_host(st) == 0).all()
    assert np.abs(_host(w) - ref).max() / np.abs(ref).max() < 1e-10
    wb = _host(engine.ban(_dev(ref), _dev(Pn)))
    refb = ob.ban(ref, Pn)
    assert np.abs(wb - refb).max() / np.abs(refb).max() < 1e-12
    zero = np.zeros_like(ref)             # denominator == 0 -> 0 (beamformer.py:483-487)
    assert (_host(engine.ban(_dev(zero), _dev(Pn))) == 0).all()
    x = _rand_c(rng, N, D, T).astype(np.complex64)
    s = _host(engine.apply_bf(_dev(ref), _dev(x)))
    refs = ob.apply_bf(ref, x.astype(np.complex128))
    assert np.abs(s - refs).max() / np.abs(refs).max() < 1e-13
    mat, num, den, st = engine.mvdr_souden(_dev(Pt), _dev(Pn), np.finfo(np.float64).tiny)
    mat, num, den = _host(mat), _host(num), _host(den)
    phi = np.linalg.solve(Pn, Pt)
    refm = phi / np.trace(phi, axis1=-1, axis2=-2).real[:, None, None]
    assert np.abs(mat - refm).max() / np.abs(refm).max() < 1e-10
    rnum = np.einsum('FdR,FdD,FDR->FR', refm.conj(), Pt, refm)
    rden = np.einsum('FdR,FdD,FDR->FR', refm.conj(), Pn, refm)
    assert np.abs(num - rnum).max() / np.abs(rnum).max() < 1e-10
    assert np.abs(den - rden).max() / np.abs(rden).max() < 1e-10
