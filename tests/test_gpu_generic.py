"""GPU: the generic-size path (9 <= D <= 32 sensors, csrc/generic.hip) of the cACGMM trainer,
predict, M-step, eigendecomposition and PSD against the NumPy oracle.  Same protocol and
tolerances as tests/test_gpu_em.py (single step 1e-9, trajectories well inside 1e-5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cov(vec, val):
    return np.einsum('...wx,...x,...zx->...wz', vec, val, vec.conj())


@pytest.mark.parametrize('D', [9, 12, 16, 17, 24, 32])
def test_heev_matches_eigh(D):
    from pb_bss_amd import _lib, engine
    rng = np.random.default_rng(D)
    a = rng.standard_normal((7, D, D)) + 1j * rng.standard_normal((7, D, D))
    a = a @ a.conj().swapaxes(-1, -2) + np.eye(D) * rng.uniform(0, 2, size=(7, 1, 1))
    a[3] = np.diag(np.arange(D, 0, -1.0))       # already diagonal, descending
    a[5, :, :] = 0                               # zero matrix
    val, vec, st = engine.heev(_lib.to_device(a))
    val, vec = _lib.to_host(val), _lib.to_host(vec)
    ref = np.linalg.eigvalsh(a)
    np.testing.assert_allclose(val, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(_cov(vec, val), a, atol=1e-11 * np.abs(a).max())
    eye = np.einsum('ndk,ndl->nkl', vec.conj(), vec)
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(D), eye.shape), atol=1e-12)
    assert int(_lib.to_host(st).max()) == 0


@pytest.mark.parametrize('F,T,D,K', [(5, 70, 9, 2), (4, 300, 16, 3), (3, 130, 24, 5), (2, 90, 32, 6)])
def test_single_step_and_trajectory(F, T, D, K):
    from oracle import cacgmm as oc, synth
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(F, T, D, K, seed=D + K)
    Y128 = Y.astype(np.complex128)
    for iters, tol in ((1, 1e-9), (8, 1e-7)):
        m = oc.em_fit(Y128, init, iterations=iters)
        model = CACGMMTrainer().fit(Y, initialization=init, iterations=iters)
        assert model.cacg.covariance_eigenvectors.shape == (F, K, D, D)
        np.testing.assert_allclose(model.weight, m['weight'], atol=tol)
        np.testing.assert_allclose(model.cacg.covariance, _cov(m['eigvec'], m['eigval']), atol=tol)
        np.testing.assert_allclose(model.predict(Y), oc.em_predict(m, Y128), atol=tol)


@pytest.mark.parametrize('F,T,D,K', [(4, 200, 6, 7), (3, 260, 12, 9), (2, 300, 9, 16), (3, 150, 24, 8)])
def test_many_classes(F, T, D, K):
    """7 <= K <= 16 classes run on the generic path at any D (class chunks of <= 6 in gen_cov,
    LDS softmax in gen_estep)."""
    from oracle import beamformer as ob, cacgmm as oc, synth
    from pb_bss_amd import extraction as ex
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(F, T, D, K, seed=D * K)
    Y128 = Y.astype(np.complex128)
    sal = np.random.default_rng(1).uniform(0.3, 1.0, size=(F, T))
    for kw in ({}, dict(saliency=sal), dict(weight_constant_axis=-2)):
        m = oc.em_fit(Y128, init, iterations=5, **kw)
        model = CACGMMTrainer().fit(Y, initialization=init, iterations=5, **kw)
        np.testing.assert_allclose(model.weight, m['weight'], atol=1e-8)
        np.testing.assert_allclose(model.predict(Y), oc.em_predict(m, Y128), atol=1e-7)
    masks = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=5)
    np.testing.assert_allclose(
        masks, CACGMMTrainer().fit(Y, initialization=init, iterations=5).predict(Y), atol=1e-10)
    X = np.ascontiguousarray(Y.transpose(0, 2, 1))
    np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X, masks),
                               ob.psd(X.astype(np.complex128), masks), atol=1e-11)


def test_guided_source_separation_shape():
    """The GSS recipe: many channels, speakers + noise classes, a boolean source activity mask
    from the diarisation as both initialisation support and E-step constraint
    (cacgmm.py:150-170, 242-247)."""
    from oracle import cacgmm as oc, synth
    from pb_bss_amd.distribution import CACGMMTrainer
    F, T, D, K = 3, 240, 24, 9
    Y, _ = synth.make_stft(F, T, D, K, seed=3)
    Y128 = Y.astype(np.complex128)
    rng = np.random.default_rng(2)
    act = np.zeros((K, T), dtype=bool)
    act[-1] = True                                             # noise class always active
    for k in range(K - 1):
        a0 = rng.integers(0, T - 60)
        act[k, a0:a0 + rng.integers(30, 60)] = True
    act = np.broadcast_to(act, (F, K, T)).copy()
    init = act / act.sum(axis=-2, keepdims=True)
    m = oc.em_fit(Y128, init, iterations=6, source_activity_mask=act)
    model = CACGMMTrainer().fit(Y, initialization=init, iterations=6, source_activity_mask=act)
    ref = oc.em_predict(m, Y128, source_activity_mask=act)
    got = model.predict(Y, source_activity_mask=act)
    np.testing.assert_allclose(got, ref, atol=1e-7)
    assert np.all(got[~act] == 0)


def test_options_resume_and_complex128_input():
    from oracle import cacgmm as oc, synth
    from pb_bss_amd.distribution import CACGMMTrainer
    F, T, D, K = 4, 120, 12, 3
    Y, init = synth.make_stft(F, T, D, K, seed=5)
    Y128 = Y.astype(np.complex128)
    Y128[1, 7] = 0                                                   # an all-zero frame
    sal = np.random.default_rng(0).uniform(0.3, 1.0, size=(F, T))
    for kw in (dict(covariance_norm='trace'), dict(covariance_norm=False, eigenvalue_floor=1e-6),
               dict(weight_constant_axis=-2), dict(saliency=sal)):
        okw = dict(kw)
        m = oc.em_fit(Y128, init, iterations=4, **okw)
        model = CACGMMTrainer().fit(Y128, initialization=init, iterations=4, **kw)
        np.testing.assert_allclose(model.predict(Y128), oc.em_predict(m, Y128), atol=1e-8)
    # resume from a model == run straight through
    a = CACGMMTrainer().fit(Y, initialization=init, iterations=3)
    b = CACGMMTrainer().fit(Y, initialization=a, iterations=2)
    c = CACGMMTrainer().fit(Y, initialization=init, iterations=5)
    np.testing.assert_allclose(b.predict(Y), c.predict(Y), atol=1e-9)
    masks = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=5)
    np.testing.assert_allclose(masks, c.predict(Y), atol=1e-10)


def test_psd_and_pca_for_many_sensors():
    from oracle import beamformer as ob
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(2)
    F, D, K, T = 6, 20, 3, 150
    X = (rng.standard_normal((F, D, T)) + 1j * rng.standard_normal((F, D, T))).astype(np.complex64)
    mask = rng.uniform(size=(F, K, T))
    X128 = X.astype(np.complex128)
    np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X, mask), ob.psd(X128, mask),
                               atol=1e-12)
    np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X, mask, normalize=False),
                               ob.psd(X128, mask, normalize=False), atol=1e-10)
    np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X), ob.psd(X128), atol=1e-12)
    psd = ob.psd(X128, mask)
    w = ex.get_pca_vector(psd[:, 0])
    ref = ob.pca_vector(psd[:, 0])
    cs = np.abs(np.einsum('fd,fd->f', w.conj(), ref))
    np.testing.assert_allclose(cs, 1.0, atol=1e-10)
    s = ex.apply_beamforming_vector(w, X)
    np.testing.assert_allclose(s, ob.apply_bf(w, X128), atol=1e-10)


def _psd(rng, F, D, rank=None):
    r = 2 * D if rank is None else rank
    a = rng.standard_normal((F, D, r)) + 1j * rng.standard_normal((F, D, r))
    return a @ a.conj().swapaxes(-1, -2) / r


def _cos(a, b):
    return np.abs(np.einsum('...d,...d->...', a.conj(), b)) / (
        np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize('D', [9, 16, 24, 32])
def test_beamformers_for_many_sensors(D):
    from oracle import beamformer as ob
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(D)
    F = 11
    target, noise = _psd(rng, F, D), _psd(rng, F, D) + 0.05 * np.eye(D)
    # GEV: same eigenvector up to phase, normalised w^H N w = 1
    w = ex.get_gev_vector(target, noise)
    np.testing.assert_allclose(_cos(w, ob.gev_vector(target, noise)), 1.0, atol=1e-9)
    np.testing.assert_allclose(np.einsum('fd,fde,fe->f', w.conj(), noise, w).real, 1.0, atol=1e-9)
    # MVDR-Souden with automatic and fixed reference channel, wMWF, MVDR, BAN, stable_solve
    np.testing.assert_allclose(ex.get_mvdr_vector_souden(target, noise),
                               ob.mvdr_souden(target, noise), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ex.get_mvdr_vector_souden(target, noise, ref_channel=D - 1),
                               ob.mvdr_souden(target, noise, ref_channel=D - 1), rtol=1e-8,
                               atol=1e-10)
    np.testing.assert_allclose(ex.get_wmwf_vector(target, noise, reference_channel=0),
                               ob.wmwf(target, noise, reference_channel=0), rtol=1e-8, atol=1e-10)
    atf = ob.pca_vector(target)
    np.testing.assert_allclose(ex.get_mvdr_vector(atf, noise), ob.mvdr(atf, noise), rtol=1e-8,
                               atol=1e-10)
    np.testing.assert_allclose(ex.blind_analytic_normalization(w, noise), ob.ban(w, noise),
                               rtol=1e-10, atol=1e-12)
    B = rng.standard_normal((F, D, 3)) + 1j * rng.standard_normal((F, D, 3))
    np.testing.assert_allclose(ex.stable_solve(noise, B), np.linalg.solve(noise, B), rtol=1e-8,
                               atol=1e-10)
    # the dispatcher end to end
    wb = ex.get_bf_vector('gev+ban', target, noise)
    np.testing.assert_allclose(_cos(wb, ob.bf_vector('gev+ban', target, noise)), 1.0, atol=1e-9)


def test_singular_noise_takes_least_squares_branch_for_many_sensors():
    from oracle import beamformer as ob
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(3)
    F, D = 5, 12
    target = _psd(rng, F, D)
    noise = _psd(rng, F, D)
    noise[2] = 0                                  # silent bin: exactly singular
    B = rng.standard_normal((F, D, 2)) + 1j * rng.standard_normal((F, D, 2))
    got = ex.stable_solve(noise, B)
    np.testing.assert_allclose(got, ob.stable_solve(noise, B), rtol=1e-8, atol=1e-10)
    assert np.abs(got[2]).max() == 0.0            # lstsq of a zero system: minimum norm = 0
    w = ex.get_mvdr_vector_souden(target, noise, ref_channel=1)
    np.testing.assert_allclose(w, ob.mvdr_souden(target, noise, ref_channel=1), rtol=1e-8,
                               atol=1e-10)
    # GEV on a non-positive-definite noise matrix: ValueError like the Cython path
    with pytest.raises(ValueError):
        ex.get_gev_vector(target, noise)
