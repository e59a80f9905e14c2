"""GPU: the generic-size path (9 <= D <= 32 sensors, csrc/generic.hip) of the cACGMM trainer,
predict, M-step, eigendecomposition and PSD against the NumPy oracle.  Same protocol and
tolerances as tests/test_gpu_em.py (single step 1e-9, trajectories well inside 1e-5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cov(vec, val):
    return np.einsum('...wx,...x,...zx->...wz', vec, val, vec.conj())


@pytest.mark.parametrize('D', [9, 12, 16, 17, 24, 32, 33, 34])
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


@pytest.mark.parametrize('F,T,D,K', [(5, 70, 9, 2), (4, 300, 16, 3), (3, 130, 24, 5), (2, 90, 32, 6),
                                     # the sensor counts between the engine's 32 x 32 tiles and the
                                     # reference's `assert D < 35` (cacgmm.py:250): 36-wide tiles
                                     (3, 140, 33, 2), (2, 260, 34, 3)])
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


def test_remainder_bins_run_as_a_side_chain():
    """2^n + 1 bins: the bins beyond a multiple of the CU count run as their own chain of launches
    on the side stream (pbbss_cacgmm_fit, generic-size path); same results as one chain
    (pbbss_set_split_tail(0)) and as the oracle, with saliency, activity mask and a model resume."""
    from oracle import cacgmm as oc, synth
    from pb_bss_amd import engine
    from pb_bss_amd.distribution import CACGMMTrainer
    F, T, D, K = 257 + 2, 60, 9, 2
    Y, init = synth.make_stft(F, T, D, K, seed=7)
    Y128 = Y.astype(np.complex128)
    sal = np.random.default_rng(0).uniform(0.2, 1.0, size=(F, T))
    kw = dict(iterations=4, saliency=sal)
    ref = oc.em_fit(Y128, init, **kw)
    got = {}
    try:
        for split in (1, 0):
            engine.set_split_tail(split)
            model = CACGMMTrainer().fit(Y, initialization=init, **kw)
            got[split] = (model.weight, model.cacg.covariance, model.predict(Y))
            resumed = CACGMMTrainer().fit(Y, initialization=model, iterations=2, saliency=sal)
            got[split] += (resumed.predict(Y),)
    finally:
        engine.set_split_tail(1)
    for a, b in zip(got[1], got[0]):
        assert np.array_equal(a, b)  # same kernels on the same bins: bit-identical
    np.testing.assert_allclose(got[1][0], ref['weight'], atol=1e-8)
    np.testing.assert_allclose(got[1][1], _cov(ref['eigvec'], ref['eigval']), atol=1e-8)
    np.testing.assert_allclose(got[1][2], oc.em_predict(ref, Y128), atol=1e-8)


@pytest.mark.parametrize('F,T,D,K', [(4, 200, 6, 7), (3, 260, 12, 9), (2, 300, 9, 16), (3, 150, 24, 8),
                                     (2, 260, 9, 19), (2, 200, 8, 17)])
def test_many_classes(F, T, D, K):
    """7 <= K <= 19 classes (the reference asserts K < 20, cacgmm.py:249) run on the generic path at any D (class chunks of <= 6 in gen_cov,
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


def test_sensor_counts_33_and_34_are_served_35_is_the_references_assert():
    """The reference's sanity assert admits D < 35 (cacgmm.py:250).  Rounds 1-5 stopped at the
    32 x 32 tiles of the generic-size kernels; D = 33, 34 now run on 36-wide tiles (E-step,
    covariance, QL eigensolver, Gauss-Jordan fast path): a full fit with saliency, the predict of
    the fitted model and fit_predict against the oracle; D >= 35 is the reference's AssertionError."""
    from oracle import cacgmm as oc, synth
    from pb_bss_amd.distribution import CACGMMTrainer
    rng = np.random.default_rng(0)
    for D, K in ((33, 3), (34, 2)):
        F, T = 3, 420   # >= 4 D frames per class: the covariances stay well conditioned
        Y, init = synth.make_stft(F, T, D, K, seed=D)
        sal = rng.uniform(0.2, 1.0, size=(F, T))
        Y128 = Y.astype(np.complex128)
        ref = oc.em_fit(Y128, init, iterations=8, saliency=sal)
        model = CACGMMTrainer().fit(Y, initialization=init, iterations=8, saliency=sal)
        np.testing.assert_allclose(model.weight, ref['weight'], atol=1e-8)
        np.testing.assert_allclose(model.cacg.covariance, _cov(ref['eigvec'], ref['eigval']), atol=1e-7)
        want = oc.em_predict(ref, Y128)
        np.testing.assert_allclose(model.predict(Y), want, atol=1e-7)
        np.testing.assert_allclose(
            CACGMMTrainer().fit_predict(Y, initialization=init, iterations=8, saliency=sal), want,
            atol=1e-7)
    Y = (rng.standard_normal((2, 40, 35)) + 1j * rng.standard_normal((2, 40, 35))).astype(np.complex64)
    init = rng.uniform(size=(2, 2, 40))
    with pytest.raises(AssertionError, match='Channels'):
        CACGMMTrainer().fit(Y, initialization=init / init.sum(1, keepdims=True), iterations=2)


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


def _cos_sim(a, b):
    num = np.abs(np.einsum('...d,...d', a.conj(), b))
    return num / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize('F,T,D,K,with_sal', [
    (6, 300, 12, 8, False),   # D = 12, K = 8
    (3, 200, 29, 3, True),
    (5, 257, 6, 7, False),    # few sensors, more classes than the fused kernel takes
    (4, 150, 16, 2, True),
    (2, 130, 32, 5, False),
])
def test_watson_mixture_at_generic_sizes(F, T, D, K, with_sal):
    """CWMMTrainer / CWMM.predict beyond the fused kernel's D <= 8, K <= 4
    (csrc/generic_watson.hip) against the oracle's reference loop."""
    from pb_bss_amd.distribution import CWMMTrainer
    from oracle import cwmm as ow, synth
    Y, init = synth.make_stft(F, T, D, K, seed=D * 10 + K)
    Y128 = Y.astype(np.complex128)
    sal = np.random.default_rng(D).uniform(0.2, 1.0, size=(F, T)) if with_sal else None
    for iterations in (1, 5):
        model = CWMMTrainer().fit(Y, initialization=init, iterations=iterations, saliency=sal)
        ref = ow.cwmm_fit(Y128, init, iterations=iterations, saliency=sal)
        tol = 1e-9 if iterations == 1 else 1e-6
        assert model.complex_watson.mode.shape == (F, K, D)
        np.testing.assert_allclose(model.complex_watson.concentration, ref['concentration'],
                                   rtol=tol, atol=tol)
        np.testing.assert_allclose(model.weight, ref['weight'], atol=tol)
        assert np.abs(_cos_sim(model.complex_watson.mode, ref['mode']) - 1).max() < tol
        aff = model.predict(Y)
        assert aff.shape == (F, K, T)
        assert np.abs(aff - ow.cwmm_predict(ref, Y128)).max() < 100 * tol
    # bins coupled through the weights: the step-wise loop on the same kernels
    m2 = CWMMTrainer().fit(Y, initialization=init, iterations=3, weight_constant_axis=(-3, -1))
    r2 = ow.cwmm_fit(Y128, init, iterations=3, weight_constant_axis=(-3, -1))
    assert np.abs(m2.predict(Y) - ow.cwmm_predict(r2, Y128)).max() < 1e-6


def test_watson_log_norm_for_many_sensors():
    """ln c(kappa) of the generic path (series of 1F1(1; D; kappa), all terms positive) against
    scipy over the whole range of the concentration spline."""
    from pb_bss_amd import _lib
    from pb_bss_amd.distribution import CWMM, ComplexWatson
    from oracle import cwmm as ow
    rng = np.random.default_rng(0)
    for D in (9, 16, 32):
        conc = np.concatenate([[0.0, 1e-3, 0.5, 7.9, 8.0, 60.0, 499.0, 500.0],
                               rng.uniform(0, 500, size=8)])
        K = conc.size
        mode = rng.standard_normal((K, D)) + 1j * rng.standard_normal((K, D))
        mode /= np.linalg.norm(mode, axis=-1, keepdims=True)
        y = rng.standard_normal((1, 40, D)) + 1j * rng.standard_normal((1, 40, D))
        model = CWMM(weight=np.full((K, 1), 1.0 / K),
                     complex_watson=ComplexWatson(mode=mode, concentration=conc))
        aff = model.predict(y)
        want = ow.cwmm_predict(dict(weight=model.weight, mode=mode, concentration=conc), y)
        assert np.abs(aff - want).max() < 1e-9, D


@pytest.mark.parametrize('kind,D,K,kw', [
    ('gaussian', 12, 6, {}),
    ('gaussian', 16, 3, dict(weight_constant_axis=(-3, -1), spectral_weight=0.5)),
    ('gaussian', 9, 4, dict(weight_constant_axis=(-1,))),
    ('gaussian', 24, 2, dict(covariance_type='full')),
    ('vmf', 12, 3, dict(weight_constant_axis=(-3, -1), max_concentration=80.)),
    ('gaussian', 12, 8, {}),                                  # D = 12, K = 8
    ('gaussian', 6, 7, dict(weight_constant_axis=(-3,))),     # few sensors, 7 classes
    ('vmf', 5, 8, dict(max_concentration=80.)),
    ('gaussian', 10, 2, dict(weight_constant_axis=(-2,))),      # uniform class weights
    ('gaussian', 11, 3, dict(weight_constant_axis=(-3, -2, -1), covariance_type='diagonal')),
    ('gaussian', 12, 3, dict(weight_constant_axis=(-3,), inline_permutation_alignment=True)),
    ('vmf', 9, 4, dict(weight_constant_axis=(-3, -1), inline_permutation_alignment=True,
                       max_concentration=80.)),
])
def test_joint_models_at_generic_sizes(kind, D, K, kw):
    """GCACGMM / VMFCACGMM with more than 8 sensors: the spatial half on the generic-size kernels
    (E-step with the spectral log-pdf as extra exponent, covariance, eigh), same spectral kernels."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, E = 7, 260, 20
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=D + K)
    sal = np.random.default_rng(1).uniform(0.2, 1.0, size=(F, T))
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    for iterations in (2, 6):
        model = trainer.fit(Y, e, initialization=init, iterations=iterations, saliency=sal, **kw)
        ref = oe.joint_fit(kind, Y128, e64, init, iterations, saliency=sal, **kw)
        want = oe.joint_model_predict(ref, Y128, e64)
        masks = model.predict(Y, e)
        assert masks.shape == (F, K, T)
        assert np.abs(masks - want).max() < (1e-8 if iterations == 2 else 1e-6), iterations
