"""GPU: behaviour tests of the drop-in API, modelled on the reference's own
test-suite (tests/test_distribution/test_cacgmm.py, tests/test_extraction/*)."""
import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sample_cacg(rng, size, covariance):
    """Unit-norm samples of a complex Gaussian (what sample_cacgmm draws,
    distribution/cacgmm.py:27-55)."""
    D = covariance.shape[-1]
    L = np.linalg.cholesky(covariance)
    z = (rng.standard_normal((size, D)) + 1j * rng.standard_normal((size, D))) / np.sqrt(2)
    x = z @ L.T
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def _sample_cacgmm(rng, size, weight, covariance):
    labels = rng.choice(len(weight), size=size, p=weight)
    x = np.zeros((size, covariance.shape[-1]), np.complex128)
    for k in range(len(weight)):
        x[labels == k] = _sample_cacg(rng, int((labels == k).sum()), covariance[k])
    return x


def _best_perm(a, b):
    return min(itertools.permutations(range(len(a))),
               key=lambda p: np.abs(a[list(p)] - b).sum())


def test_cacgmm_recovers_parameters():
    """tests/test_distribution/test_cacgmm.py:25-53: 10000 samples, D=3, K=2,
    covariance atol 0.1, weights atol 0.15."""
    from pb_bss_amd.distribution import CACGMMTrainer
    rng = np.random.default_rng(0)
    np.random.seed(0)
    D = 3
    cov1 = np.array([[10, 1 + 1j, 1 + 1j], [1 - 1j, 5, 1], [1 - 1j, 1, 2]], np.complex128)
    cov2 = np.array([[2, 0, 0], [0, 3, 0], [0, 0, 2]], np.complex128)
    cov1 /= np.trace(cov1)
    cov2 /= np.trace(cov2)
    cov = np.stack([cov1, cov2])
    weight = np.array([0.3, 0.7])
    x = _sample_cacgmm(rng, 10000, weight, cov)
    model = CACGMMTrainer().fit(x, num_classes=2, covariance_norm='trace')
    perm = list(_best_perm(model.cacg.covariance, cov))
    assert np.allclose(model.cacg.covariance[perm], cov, atol=0.1)
    assert np.allclose(model.weight[perm, 0], weight, atol=0.15)
    assert model.weight.shape == (2, 1)
    assert model.cacg.covariance_eigenvectors.shape == (2, 3, 3)
    assert model.predict(x).shape == (2, 10000)


def test_independent_dims_and_init_shapes():
    """tests/test_distribution/test_cacgmm.py:71-163: independent axes and
    broadcastable initialisations."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from oracle import cacgmm as oc
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 200, 4)) + 1j * rng.standard_normal((3, 200, 4))
    init = rng.uniform(size=(1, 2, 200))
    init /= init.sum(-2, keepdims=True)
    m = CACGMMTrainer().fit(x, initialization=init, iterations=3)
    assert m.weight.shape == (3, 2, 1)
    ref = oc.em_fit(x, np.broadcast_to(init, (3, 2, 200)), iterations=3)
    assert np.abs(m.weight - ref['weight']).max() < 1e-10
    with pytest.raises(AssertionError):
        CACGMMTrainer().fit(x, initialization=init[0], iterations=1)  # wrong ndim
    with pytest.raises(AssertionError):
        CACGMMTrainer().fit(x, iterations=1)  # neither init nor num_classes
    with pytest.raises(AssertionError):
        CACGMMTrainer().fit(x.real, num_classes=2, iterations=1)  # not complex
    np.random.seed(3)
    m1 = CACGMMTrainer().fit(x, num_classes=3, iterations=2)
    np.random.seed(3)
    m2 = CACGMMTrainer().fit(x, num_classes=3, iterations=2)
    assert (m1.weight == m2.weight).all(), 'same global-RNG initialisation as the reference'
    assert m1.weight.shape == (3, 3, 1)


def test_resume_and_fit_predict_and_torch_io():
    import torch
    from pb_bss_amd.distribution import CACGMM, CACGMMTrainer
    from oracle import synth
    Y, init = synth.make_stft(6, 100, 4, 2, seed=4)
    a = CACGMMTrainer().fit(Y, initialization=init, iterations=5)
    b = CACGMMTrainer().fit(Y, initialization=init, iterations=3)
    b = CACGMMTrainer().fit(Y, initialization=b, iterations=2)
    assert isinstance(b, CACGMM)
    assert np.abs(a.predict(Y) - b.predict(Y)).max() < 1e-9
    fp = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=5)
    assert np.abs(fp - a.predict(Y)).max() < 1e-12
    yt = torch.from_numpy(Y).cuda()
    mt = CACGMMTrainer().fit(yt, initialization=torch.from_numpy(init).cuda(), iterations=5)
    assert mt.weight.is_cuda and mt.cacg.covariance_eigenvectors.is_cuda
    pt = mt.predict(yt)
    assert pt.is_cuda and np.abs(pt.cpu().numpy() - a.predict(Y)).max() < 1e-12


def test_stepwise_path_equals_fused_path():
    """weight_constant_axis=(-1,) through the per-iteration E/M entry points
    (forced with a no-op aligner-free stepwise call) equals the fused kernel."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.distribution.cacgmm import CACGMMTrainer as T
    from pb_bss_amd import _lib
    from oracle import synth
    Y, init = synth.make_stft(5, 80, 4, 3, seed=8)
    a = CACGMMTrainer().fit(Y, initialization=init, iterations=4)
    t = _lib.torch()
    yb = _lib.to_device(Y)
    b = T()._fit_stepwise(yb, (5,), 3, _lib.to_device(init), None, 4, None, None, None,
                          (-1,), 'eigenvalue', 1e-10, 1e-10, True, None, False)
    assert np.abs(a.weight - b.weight).max() < 1e-10
    assert np.abs(a.cacg.covariance - b.cacg.covariance).max() < 1e-9


def test_inline_permutation_aligner_hook():
    from pb_bss_amd.distribution import CACGMMTrainer
    from oracle import synth

    class Swap:  # swaps classes 0/1 in every second frequency
        def calculate_mapping(self, mask):
            K, F, T = mask.shape
            m = np.tile(np.arange(K)[:, None], (1, F))
            m[:2, 1::2] = m[:2, 1::2][::-1]
            return m

        @staticmethod
        def apply_mapping(mask, mapping):
            return mask[mapping, range(mapping.shape[1])]

    Y, init = synth.make_stft(4, 60, 3, 2, seed=2)
    m = CACGMMTrainer().fit(Y, initialization=init, iterations=3,
                            weight_constant_axis=(-3,), inline_permutation_aligner=Swap())
    assert m.weight.shape == (1, 2, 60)
    assert np.isfinite(m.cacg.covariance).all()


def test_unsupported_shapes_fail_loudly():
    from pb_bss_amd.distribution import CACGMMTrainer
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 50, 35)) + 1j * rng.standard_normal((2, 50, 35))
    with pytest.raises(AssertionError):
        CACGMMTrainer().fit(x, num_classes=2, iterations=1)  # the reference's `assert D < 35`
    x = rng.standard_normal((2, 50, 4)) + 1j * rng.standard_normal((2, 50, 4))
    m = CACGMMTrainer().fit(x, num_classes=19, iterations=1)  # K = 17 .. 19: served since round 5
    assert m.weight.shape == (2, 19, 1)
    with pytest.raises(AssertionError, match='num_classes'):
        CACGMMTrainer().fit(x, num_classes=20, iterations=1)  # the reference's assert (cacgmm.py:249)


# ------------------------------------------------------------------ extraction
def _psd(rng, shape):
    D = shape[-1]
    x = rng.standard_normal((*shape[:-2], D, D + 3)) + 1j * rng.standard_normal((*shape[:-2], D, D + 3))
    return x @ x.conj().swapaxes(-1, -2) + 2 * D * np.eye(D)


@pytest.mark.parametrize('shape', [(51, 6, 6), (1, 6, 6), (2, 51, 6, 6)])
def test_beamformer_dimensions(shape):
    """tests/test_extraction/test_beamformer.py:25-118: three shape regimes."""
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(0)
    t, n = _psd(rng, shape), _psd(rng, shape)
    assert ex.get_gev_vector(t, n).shape == shape[:-1]
    assert ex.get_pca_vector(t).shape == shape[:-1]
    assert ex.get_mvdr_vector_souden(t, n, ref_channel=0).shape == shape[:-1]
    if len(shape) == 3:
        assert ex.get_mvdr_vector_souden(t, n).shape == shape[:-1]
    else:
        with pytest.raises(ValueError):
            ex.get_mvdr_vector_souden(t, n)
    w = ex.get_gev_vector(t, n)
    assert ex.blind_analytic_normalization(w, n).shape == shape[:-1]
    x = rng.standard_normal((*shape[:-2], 6, 40)) + 0j
    assert ex.apply_beamforming_vector(w, x).shape == (*shape[:-2], 40)
    assert ex.get_mvdr_vector(w, n).shape == shape[:-1]


def test_gev_equals_pca_for_identity_noise_and_wrapper_equivalences():
    """test_beamformer.py:98-104 and test_beamformer_wrapper.py:39-91."""
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(1)
    t = _psd(rng, (33, 6, 6))
    n = _psd(rng, (33, 6, 6))
    eye = np.broadcast_to(np.eye(6), t.shape)

    def cos(a, b):
        return np.abs(np.einsum('...d,...d', a.conj(), b)) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)

    assert np.abs(cos(ex.get_gev_vector(t, eye), ex.get_pca_vector(t)) - 1).max() < 1e-6
    for name in ['pca', 'pca+mvdr', 'scaled_gev_atf+mvdr', 'mvdr_souden', 'gev',
                 'rank1_pca+mvdr_souden', 'rank1_gev+mvdr_souden', 'rank1_pca+gev',
                 'rank1_gev+gev', 'gev+ban', 'mvdr_souden+ban', 'ch0']:
        assert ex.get_bf_vector(name, t, n).shape == (33, 6), name
    assert np.abs(cos(ex.get_bf_vector('rank1_gev+gev', t, n), ex.get_bf_vector('gev', t, n)) - 1).max() < 1e-6
    with pytest.raises(ValueError):
        ex.get_bf_vector('nonsense', t, n)
    with pytest.raises(AssertionError):
        ex.get_bf_vector('lcmv', t, n)


def test_gev_not_positive_definite_raises():
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(2)
    t = _psd(rng, (5, 4, 4))
    n = _psd(rng, (5, 4, 4))
    n[3] = -n[3]
    with pytest.raises(np.linalg.LinAlgError):
        ex.get_gev_vector(t, n)
    with pytest.raises(ValueError, match='not positive'):
        ex.get_gev_vector(t, n, force_cython=True)


def test_mvdr_souden_difficulties():
    """tests/test_extraction/test_beamformer.py:211-376: zero / inf inputs."""
    from pb_bss_amd.extraction import get_mvdr_vector_souden
    obs = np.array([[0, 0, 1], [0, 0.1, 1], [0.1, 0, 1]])
    pxx = obs.T.conj() @ obs
    pnn = np.eye(3)
    well, = get_mvdr_vector_souden(pxx[None], pnn[None])
    w3 = get_mvdr_vector_souden([pxx] * 3, [pnn] * 3)
    assert np.allclose([well] * 3, w3)
    for a, b in [(pxx[None] * 0, pnn[None]), (pxx[None], pnn[None] * 0),
                 (pxx[None] * 0, pnn[None] * 0)]:
        w = get_mvdr_vector_souden(a, b)
        assert (w == 0).all(), w
        with pytest.raises(AssertionError):
            get_mvdr_vector_souden(a, b, eps=0)
    with np.errstate(all='ignore'):
        for a, b in [(pxx[None] * np.inf, pnn[None]), (pxx[None], pnn[None] * np.inf)]:
            with pytest.raises(AssertionError):
                get_mvdr_vector_souden(a, b)
    for a, b in [([pxx * 0, pxx], [pnn, pnn]), ([pxx, pxx], [pnn * 0, pnn]),
                 ([pxx * 0, pxx], [pnn * 0, pnn])]:
        w, ref = get_mvdr_vector_souden(a, b, return_ref_channel=True)
        assert ref == 2
        assert np.allclose(w, np.array([[0., 0., 0.], well]))


def test_psd_properties():
    """tests/test_extraction/test_covariance_matrix.py:31-155."""
    from pb_bss_amd.extraction import get_power_spectral_density_matrix as psd
    rng = np.random.default_rng(3)
    F, T, D, K = 21, 57, 5, 2
    X = rng.standard_normal((F, D, T)) + 1j * rng.standard_normal((F, D, T))
    mask = rng.uniform(size=(F, K, T))
    p = psd(X, mask)
    assert p.shape == (F, K, D, D)
    assert np.abs(p - p.conj().swapaxes(-1, -2)).max() < 1e-14  # Hermitian
    assert (np.linalg.eigvalsh(p) > -1e-12).all()                # PSD
    assert np.abs(psd(X, 7.5 * mask) - p).max() < 1e-13          # mask scale invariance
    assert np.abs(psd(X, mask > 0.5) - psd(X, (mask > 0.5).astype(float))).max() == 0  # bool mask
    assert np.abs(psd(X.transpose(0, 2, 1), mask.transpose(0, 2, 1), sensor_dim=-1,
                      source_dim=-1, time_dim=-2) - p).max() < 1e-14


@pytest.mark.parametrize('K,D', [(5, 6), (6, 8), (5, 3)])
def test_more_classes(K, D):
    """K = 5, 6 run on the 2-workgroups-per-CU instantiation."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from oracle import cacgmm as oc, synth
    Y, init = synth.make_stft(6, 150, D, K, seed=K * 10 + D)
    m = CACGMMTrainer().fit(Y, initialization=init, iterations=5)
    Y128 = Y.astype(np.complex128)
    ref = oc.em_fit(Y128, init, iterations=5)
    assert np.abs(m.weight - ref['weight']).max() < 1e-10
    assert np.abs(m.predict(Y) - oc.em_predict(ref, Y128)).max() < 1e-8


# ---- N4: the rest of the beamformer family (bf_extra.hip) ----------------------
def _extra():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                                'beamformer_extra_f19_d5.npz'))


def test_lcmv_matches_reference_and_constraints():
    from pb_bss_amd import extraction as ex
    g = _extra()
    for resp, key in (([1, 0], 'lcmv_10'), ([0, 1], 'lcmv_01')):
        w = ex.get_lcmv_vector(g['atf'], resp, g['noise'])
        assert w.shape == g[key].shape
        np.testing.assert_allclose(w, g[key], rtol=1e-9, atol=1e-11)
        # the constraints H^H w = response hold per bin
        got = np.einsum('kfd,fd->fk', g['atf'].conj(), w)
        np.testing.assert_allclose(got, np.broadcast_to(np.array(resp, dtype=complex), got.shape),
                                   atol=1e-9)
    # K = D and D = 8 also run (padding-free and full-width paths)
    rng = np.random.default_rng(3)
    for K, D in ((3, 3), (2, 8), (8, 8)):
        from oracle import beamformer as ob
        atf = rng.standard_normal((K, 7, D)) + 1j * rng.standard_normal((K, 7, D))
        a = rng.standard_normal((7, D, 2 * D)) + 1j * rng.standard_normal((7, D, 2 * D))
        noise = a @ a.conj().swapaxes(-1, -2)
        resp = np.zeros(K)
        resp[K - 1] = 1
        np.testing.assert_allclose(ex.get_lcmv_vector(atf, resp, noise), ob.lcmv(atf, resp, noise),
                                   rtol=1e-7, atol=1e-9)


def test_lcmv_rank_deficient_noise_takes_least_squares_branch():
    from pb_bss_amd import extraction as ex
    g = _extra()
    w = ex.get_lcmv_vector(g['atf'], [1, 0], g['noise_sing'])
    ok = np.ones(19, dtype=bool)
    ok[4] = False
    np.testing.assert_allclose(w[ok], g['lcmv_sing'][ok], rtol=1e-9, atol=1e-11)
    assert np.isfinite(w[4]).all()


def test_remaining_beamformer_functions_match_reference():
    from pb_bss_amd import extraction as ex
    g = _extra()
    at = dict(rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(ex.get_mvdr_vector_merl(g['target'], g['noise']), g['merl'], **at)
    np.testing.assert_allclose(ex.distortionless_normalization(g['w'], g['atf'][0], g['noise']),
                               g['distortionless'], **at)
    pf = ex.mvdr_snr_postfilter(g['w'], g['target'], g['noise'])
    assert pf.shape == (19, 1)
    np.testing.assert_allclose(pf, g['postfilter'], **at)
    np.testing.assert_allclose(ex.zero_degree_normalization(g['wb'], 2), g['zero_degree'], **at)
    np.testing.assert_allclose(ex.phase_correction(g['w']), g['phase_2d'], **at)
    np.testing.assert_allclose(ex.phase_correction(g['wb']), g['phase_3d'], **at)
    np.testing.assert_allclose(ex.condition_covariance(g['target'], 0.05), g['conditioned'], **at)
    out = ex.apply_online_beamforming_vector(g['vt'], g['mix'])
    assert out.shape == (19, 40)
    np.testing.assert_allclose(out, g['online'], rtol=1e-10, atol=1e-10)
    vec, val = ex.get_pca(g['target'], return_all_vecs=True)
    np.testing.assert_allclose(val, g['pca_all_val'], rtol=1e-10)
    cs = np.abs(np.einsum('fdk,fdk->fk', vec.conj(), g['pca_all_vec']))
    np.testing.assert_allclose(cs, 1.0, atol=1e-9)
    vec1, val1 = ex.get_pca(g['target'])
    np.testing.assert_allclose(val1, g['pca_val'], rtol=1e-10)
    # doctest of phase_correction (beamformer.py:529-540): input untouched, result all ones
    w = np.array([[1, 1], [-1, -1]], dtype=np.complex128)
    np.testing.assert_allclose(ex.phase_correction(w), np.ones((2, 2)), atol=1e-14)
    np.testing.assert_allclose(ex.phase_correction([w])[0], np.ones((2, 2)), atol=1e-14)
    assert w[1, 0] == -1
    with pytest.raises(NotImplementedError):
        ex.get_lcmv_vector_souden(g['target'], g['target'], g['noise'])


# ------------------------------------------------------------------ step-wise fit on the device
def _oracle_stepwise(Y128, init, iters, weight_constant_axis, saliency=None, plan=None):
    """The reference loop (cacgmm.py:252-278) from oracle pieces, optionally with an inline
    DHTV aligner (mixture_model_utils.py:264-306)."""
    from oracle import cacgmm as oc, permutation_alignment as op
    yn = oc.normalize_observation(Y128)
    aff, q, model = init, np.ones_like(init), None
    for _ in range(iters):
        if model is not None:
            aff, q, _ = oc.e_step(yn, model['weight'], model['eigvec'], model['eigval'],
                                  affiliation_eps=1e-10)
            if plan is not None:
                kft = aff.transpose(1, 0, 2)
                mapping = op.dhtv_calculate_mapping(kft, plan)
                aff = op.apply_mapping(kft, mapping).transpose(1, 0, 2)
                q = op.apply_mapping(q.transpose(1, 0, 2), mapping).transpose(1, 0, 2)
        w, vec, lam = oc.m_step(yn, q, aff, saliency=saliency,
                                weight_constant_axis=weight_constant_axis)
        model = dict(weight=w, eigvec=vec, eigval=lam)
    return model


def _force_stepwise(monkeypatch):
    """Make pbbss_cacgmm_fit_shared 'unsupported' so that the trainer takes the step-wise loop."""
    from pb_bss_amd import engine
    monkeypatch.setattr(engine, 'em_fit_shared', lambda *a, **k: None)


@pytest.mark.parametrize('path', ['shared', 'stepwise'])
@pytest.mark.parametrize('axis,with_sal', [((-3,), False), ((-3, -1), False), ((-3,), True),
                                           ((-3, -1), True)])
def test_stepwise_fit_on_device_matches_oracle(axis, with_sal, path, monkeypatch):
    """weight_constant_axis with the frequency axis, both device paths against the reference
    loop: the cooperative kernel (pbbss_cacgmm_fit_shared: weights exchanged between the
    workgroups inside one launch) and the step-wise loop (E-step, cross-bin weight reduction
    pbbss_estimate_mixture_weight and M-step per iteration)."""
    from oracle import cacgmm as oc, synth
    from pb_bss_amd import engine
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(33, 120, 6, 3, seed=31)
    Y128 = Y.astype(np.complex128)
    sal = np.abs(Y128[..., 0]) if with_sal else None
    ref = _oracle_stepwise(Y128, init, 5, axis, saliency=sal)
    calls = []
    if path == 'stepwise':
        _force_stepwise(monkeypatch)
    else:
        real = engine.em_fit_shared
        monkeypatch.setattr(engine, 'em_fit_shared',
                            lambda *a, **k: calls.append(1) or real(*a, **k))
    m = CACGMMTrainer().fit(Y, initialization=init, iterations=5, weight_constant_axis=axis,
                            saliency=sal)
    assert path == 'stepwise' or calls  # the cooperative kernel really served the call
    assert m.weight.shape == ref['weight'].shape
    assert np.abs(m.weight - ref['weight']).max() < 1e-10
    assert np.abs(m.predict(Y) - oc.em_predict(ref, Y128)).max() < 1e-8


@pytest.mark.parametrize('axis', [(-3,), (-3, -1)])
def test_shared_weight_fit_batched_model_init_and_fit_predict(axis):
    """Cooperative kernel with two utterances in one call (two weight groups), resumed from a
    model (the weight plane of the caller drives the first E-step), and fit_predict (final
    E-step inside the launch) -- each against the reference loop run per utterance."""
    from oracle import cacgmm as oc, synth
    from pb_bss_amd.distribution import CACGMMTrainer
    data = [synth.make_stft(21, 90, 5, 2, seed=s) for s in (3, 4)]
    Y = np.stack([d[0] for d in data])
    init = np.stack([d[1] for d in data])
    Y128 = Y.astype(np.complex128)
    refs = [_oracle_stepwise(Y128[u], init[u], 3, axis) for u in range(2)]
    m = CACGMMTrainer().fit(Y, initialization=init, iterations=3, weight_constant_axis=axis)
    assert m.weight.shape == (2,) + refs[0]['weight'].shape
    for u in range(2):
        assert np.abs(m.weight[u] - refs[u]['weight']).max() < 1e-10
    # resume: 2 + 1 iterations == 3 iterations
    m2 = CACGMMTrainer().fit(Y, initialization=init, iterations=2, weight_constant_axis=axis)
    m3 = CACGMMTrainer().fit(Y, initialization=m2, iterations=1, weight_constant_axis=axis)
    assert np.abs(m3.weight - m.weight).max() < 1e-10
    aff = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=3,
                                      weight_constant_axis=axis)
    for u in range(2):
        assert np.abs(aff[u] - oc.em_predict(refs[u], Y128[u])).max() < 1e-8
        assert np.abs(m3.predict(Y)[u] - oc.em_predict(refs[u], Y128[u])).max() < 1e-8


@pytest.mark.parametrize('axis', [(-3,), (-3, -1)])
def test_shared_weight_fit_full_size_equals_stepwise(axis, monkeypatch):
    """F = 513 bins x T = 500 frames: all 513 workgroups co-resident (3 per CU); the
    cooperative launch and the step-wise loop walk the same trajectory."""
    from oracle import synth
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(513, 500, 8, 3, seed=5)
    a = CACGMMTrainer().fit(Y, initialization=init, iterations=12, weight_constant_axis=axis)
    _force_stepwise(monkeypatch)
    b = CACGMMTrainer().fit(Y, initialization=init, iterations=12, weight_constant_axis=axis)
    assert a.weight.shape == b.weight.shape
    assert np.abs(a.weight - b.weight).max() < 1e-9
    assert np.abs(a.predict(Y) - b.predict(Y)).max() < 1e-7


@pytest.mark.parametrize('axis,with_sal', [((-3,), False), ((-3, -1), True)])
def test_sharded_fit_with_weights_shared_over_the_sharded_bins(axis, with_sal):
    """weight_constant_axis containing the sharded bin axis: fit_predict_sharded runs the
    step-wise loop with the all-reduce hook between E and M.  One rank here (the hook's
    world-size-2 arithmetic is covered on CPU, tests/test_sharding_gloo.py): same masks as the
    unsharded cooperative-kernel fit."""
    import torch.distributed as dist
    from oracle import synth
    from pb_bss_amd import sharding
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(19, 100, 5, 3, seed=41)
    sal = np.abs(Y[..., 0]).astype(np.float64) if with_sal else None
    ref = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=4,
                                      weight_constant_axis=axis, saliency=sal)
    created = False
    if not dist.is_initialized():
        import os
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('gloo', rank=0, world_size=1)
        created = True
    try:
        got = sharding.fit_predict_sharded(Y, init, 4, weight_constant_axis=axis, saliency=sal)
    finally:
        if created:
            dist.destroy_process_group()
    from pb_bss_amd import _lib
    assert np.abs(_lib.to_host(got) - ref).max() < 1e-9


def test_shared_weight_fit_falls_back_when_not_served(monkeypatch):
    """K = 5 classes is outside the cooperative kernel: PBBSS_ERR_UNSUPPORTED from the C ABI,
    the trainer runs the step-wise loop (same result as forcing it)."""
    from oracle import synth
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(9, 80, 4, 5, seed=8)
    y = _lib.to_device(Y)
    assert engine.em_fit_shared(y, 5, 9, weight_mode=_lib.WEIGHT_SHARED_KT,
                                gamma0=_lib.to_device(init)) is None
    a = CACGMMTrainer().fit(Y, initialization=init, iterations=3, weight_constant_axis=(-3,))
    _force_stepwise(monkeypatch)
    b = CACGMMTrainer().fit(Y, initialization=init, iterations=3, weight_constant_axis=(-3,))
    assert np.abs(a.weight - b.weight).max() == 0.0


def test_stepwise_fit_with_device_inline_aligner_matches_oracle():
    """inline_permutation_aligner = the DEVICE DHTV solver: no host round trip in the loop; same
    trajectory as the reference loop with the (NumPy) DHTV aligner."""
    from oracle import cacgmm as oc, permutation_alignment as op, synth
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    F = 65
    Y, init = synth.make_stft(F, 150, 4, 2, seed=12)
    Y128 = Y.astype(np.complex128)
    kw = dict(stft_size=2 * (F - 1), segment_start=10, segment_width=20, segment_shift=5,
              main_iterations=8, sub_iterations=2)
    solver = DHTVPermutationAlignment(**kw)
    plan = np.asarray(solver.alignment_plan)
    ref = _oracle_stepwise(Y128, init, 4, (-3,), plan=plan)
    m = CACGMMTrainer().fit(Y, initialization=init, iterations=4, weight_constant_axis=(-3,),
                            inline_permutation_aligner=solver)
    assert np.abs(m.weight - ref['weight']).max() < 1e-9
    assert np.abs(m.predict(Y) - oc.em_predict(ref, Y128)).max() < 1e-7


def test_estimate_mixture_weight_kernel_against_numpy():
    import torch
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.distribution.mixture_model_utils import estimate_mixture_weight
    rng = np.random.default_rng(1)
    aff = rng.uniform(size=(3, 7, 4, 50))
    aff /= aff.sum(-2, keepdims=True)
    sal = rng.uniform(size=(3, 7, 50))
    sal[1, 2] = 0
    for red_inner, red_n, axes in ((True, False, (-3,)), (True, True, (-3, -1)),
                                   (False, True, (-1,))):
        for s in (None, sal):
            want = estimate_mixture_weight(aff, s, axes)
            got = engine.estimate_mixture_weight(
                _lib.to_device(aff), None if s is None else _lib.to_device(s), red_inner, red_n)
            assert tuple(got.shape) == want.shape
            assert np.abs(_lib.to_host(got) - want).max() < 1e-13


def test_mixture_weight_of_few_long_problems():
    """One problem of 256 500 samples (a mixture over an utterance's embeddings): the sum over the
    samples is cut into chunks over many workgroups; also three problems summed together."""
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.distribution.mixture_model_utils import estimate_mixture_weight
    rng = np.random.default_rng(3)
    for Bi, N in ((1, 256500), (3, 70001), (2, 900)):
        aff = rng.uniform(size=(1, Bi, 3, N))
        aff /= aff.sum(-2, keepdims=True)
        sal = rng.uniform(size=(1, Bi, N))
        for red_inner, axes in ((False, (-1,)), (True, (-3, -1))):
            for s in (None, sal):
                want = estimate_mixture_weight(aff[0], None if s is None else s[0], axes)
                got = engine.estimate_mixture_weight(
                    _lib.to_device(aff), None if s is None else _lib.to_device(s), red_inner, True)
                assert got.numel() == want.size
                assert np.abs(_lib.to_host(got).reshape(want.shape) - want).max() < 1e-12


def test_phase_correction_at_utterance_size():
    """2-D input scans along the frequency axis (one workgroup, chunked scan): 513 bins, a bin
    count that leaves the last threads without a chunk, and one bin short of a full chunk."""
    from oracle import beamformer as ob
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(5)
    for F, D in ((513, 6), (300, 8), (2, 3), (1026, 2)):
        w = rng.standard_normal((F, D)) + 1j * rng.standard_normal((F, D))
        np.testing.assert_allclose(ex.phase_correction(w), ob.phase_correction(w), rtol=1e-11,
                                   atol=1e-12)


def test_mixture_weight_at_utterance_size():
    """513 bins x 997 frames (ragged last frame tile): summed over the bins, and per bin."""
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.distribution.mixture_model_utils import estimate_mixture_weight
    rng = np.random.default_rng(2)
    aff = rng.uniform(size=(1, 513, 3, 997))
    aff /= aff.sum(-2, keepdims=True)
    sal = rng.uniform(size=(1, 513, 997))
    for red_inner, red_n, axes in ((True, False, (-3,)), (False, False, ())):
        for s in (None, sal):
            want = estimate_mixture_weight(aff[0], None if s is None else s[0], axes)
            got = engine.estimate_mixture_weight(
                _lib.to_device(aff), None if s is None else _lib.to_device(s), red_inner, red_n)
            assert got.numel() == want.size
            assert np.abs(_lib.to_host(got).reshape(want.shape) - want).max() < 1e-12


def test_result_dtype_reference_follows_the_reference_operand_rules():
    """The reference computes in the precision of its operands (cacgmm.py:226-227); the table
    below was read off the unmodified reference (complex64 observations, 2 iterations)."""
    import pb_bss_amd
    from pb_bss_amd.distribution import CACGMMTrainer
    rng = np.random.default_rng(0)
    F, T, D, K = 3, 40, 4, 2
    y64 = rng.standard_normal((F, T, D)) + 1j * rng.standard_normal((F, T, D))
    y = y64.astype(np.complex64)
    g = rng.uniform(size=(F, K, T))
    g /= g.sum(-2, keepdims=True)
    np.random.seed(0)
    wide = CACGMMTrainer().fit(y, initialization=g, iterations=2)
    assert wide.weight.dtype == np.float64 and wide.predict(y).dtype == np.float64
    with pb_bss_amd.result_dtype('reference'):
        table = {   # case: (weight, eigenvectors, eigenvalues, predict(y))
            'array64': (dict(initialization=g), 'f4', 'c8', 'f4', 'f4'),
            'array32': (dict(initialization=g.astype(np.float32)), 'f4', 'c8', 'f4', 'f4'),
            'random': (dict(num_classes=K), 'f8', 'c16', 'f8', 'f8'),
            'saliency64': (dict(initialization=g, saliency=np.ones((F, T))), 'f8', 'c16', 'f8', 'f8'),
            'uniform': (dict(initialization=g, weight_constant_axis=-2), 'f8', 'c8', 'f4', 'f4'),
            'shared': (dict(initialization=g, weight_constant_axis=(-3,)), 'f4', 'c8', 'f4', 'f4'),
        }
        for name, (kw, w, vec, val, pred) in table.items():
            m = CACGMMTrainer().fit(y, iterations=2, **kw)
            got = (m.weight.dtype, m.cacg.covariance_eigenvectors.dtype,
                   m.cacg.covariance_eigenvalues.dtype, m.predict(y).dtype)
            assert got == tuple(np.dtype(x) for x in (w, vec, val, pred)), (name, got)
            assert m.predict(y64).dtype == np.float64, name     # wide observations promote
            again = CACGMMTrainer().fit(y, initialization=m, iterations=1)
            assert again.cacg.covariance_eigenvectors.dtype == np.dtype(vec), name
            aff = CACGMMTrainer().fit_predict(y, iterations=2, **kw)
            assert aff.dtype == np.dtype(pred), (name, aff.dtype)
        m = CACGMMTrainer().fit(y, initialization=g, iterations=2)
    # rounding only: the single-precision results are the float64 ones to float32 accuracy
    np.testing.assert_allclose(m.cacg.covariance_eigenvalues, wide.cacg.covariance_eigenvalues,
                               rtol=1e-6)
    np.testing.assert_allclose(m.weight, wide.weight, rtol=1e-6)
    assert pb_bss_amd.set_result_dtype('float64') == 'float64'


def test_stepwise_weights_with_saliency_and_many_classes_fall_back_to_the_host_formula():
    """K = 17 with a saliency and bin-coupled weights: pbbss_estimate_mixture_weight does not
    serve it (PBBSS_ERR_UNSUPPORTED) -- the trainer must take the NumPy formula, not raise."""
    from oracle import synth
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.distribution.cacgmm import CACGMMTrainer
    rng = np.random.default_rng(0)
    aff = rng.uniform(size=(2, 3, 17, 40))
    aff /= aff.sum(axis=-2, keepdims=True)
    sal = rng.uniform(size=(2, 3, 40))
    assert engine.estimate_mixture_weight(_lib.to_device(aff), _lib.to_device(sal), True, True) is None
    assert CACGMMTrainer._device_weight(_lib.to_device(aff), _lib.to_device(sal), (-3, -1),
                                        (2, 3)) is None


@pytest.mark.gpu
def test_stack_parameters_batched_predict_equals_individual():
    """`stack_parameters` (pb_bss/distribution/utils.py:259-316, named in SURVEY 8b among the
    downstream calls that must keep working): per-utterance models stacked into one batched
    model; its `predict` on the stacked observations equals the individual predictions."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.distribution.utils import stack_parameters
    from pb_bss_amd.testing import synth
    data = [synth.make_stft(11, 90, 4, 2, seed=40 + u) for u in range(3)]
    models = [CACGMMTrainer().fit(Y, initialization=init, iterations=4) for Y, init in data]
    stacked = stack_parameters(models)
    assert stacked.weight.shape == (3,) + models[0].weight.shape
    got = stacked.predict(np.stack([Y for Y, _ in data]))
    for u, (Y, _) in enumerate(data):
        np.testing.assert_allclose(got[u], models[u].predict(Y), atol=1e-12)


def test_fit_with_log_likelihood_history_matches_the_oracle_trajectory():
    """Per-iteration log-likelihood (optional): one value per EM iteration,
    equal to `log_likelihood` of the oracle's model after the same number of iterations, and the
    final model equals the one of a single fused fit."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.testing import synth
    from oracle import cacgmm as oc
    Y, init = synth.make_stft(9, 150, 5, 3, seed=77)
    Y128 = Y.astype(np.complex128)
    model, hist = CACGMMTrainer().fit_with_log_likelihood(Y, initialization=init, iterations=5)
    assert hist.shape == (5,)
    for n in (1, 3, 5):
        want = oc.log_likelihood(oc.em_fit(Y128, init, iterations=n), Y128)
        assert abs(hist[n - 1] - want) < 1e-7 * abs(want), (n, hist[n - 1], want)
    fused = CACGMMTrainer().fit(Y, initialization=init, iterations=5)
    np.testing.assert_allclose(model.predict(Y), fused.predict(Y), atol=1e-9)
    assert isinstance(model.weight, np.ndarray)


def test_stepwise_loop_as_a_captured_graph_is_bit_identical(monkeypatch):
    """PBBSS_STEPWISE_GRAPH=1 (opt-in, round 5): after two eager iterations the step-wise EM loop
    replays ONE captured graph per iteration -- E-step, device DHTV aligner, weight reduction,
    M-step, status accumulation, the new model copied over the graph's inputs.  Same launches, same
    arguments: the model must equal the eager loop's bit for bit."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from pb_bss_amd.testing import synth
    F, T, D, K = 257, 130, 4, 2   # stft_size 512: one of the reference's two presets
    Y, init = synth.make_stft(F, T, D, K, seed=11)
    out = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('PBBSS_STEPWISE_GRAPH', flag)
        out[flag] = CACGMMTrainer().fit(
            Y, initialization=init, iterations=9, weight_constant_axis=(-3,),
            inline_permutation_aligner=DHTVPermutationAlignment.from_stft_size(2 * (F - 1)))
    a, b = out['0'], out['1']
    assert np.array_equal(a.weight, b.weight)
    assert np.array_equal(a.cacg.covariance_eigenvalues, b.cacg.covariance_eigenvalues)
    assert np.array_equal(a.cacg.covariance_eigenvectors, b.cacg.covariance_eigenvectors)
