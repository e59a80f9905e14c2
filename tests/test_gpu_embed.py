"""GPU: real-embedding mixtures: von Mises-Fisher and
spherical-Gaussian kernels, VMFMM, and the joint spatial+spectral models GCACGMM /
VMFCACGMM, against vectors of the real reference (tests/golden/embed_*.npz) and
the NumPy oracle on larger seeded problems.  Everything goes through the C ABI."""
import ast
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def cov(vec, val):
    return np.einsum('...wx,...x,...zx->...wz', vec, val, vec.conj())


def test_vmf_log_norm_series_matches_scipy_ive():
    from pb_bss_amd.distribution import VonMisesFisher
    g = load('embed_vmf_log_norm')
    ks = g['concentrations']
    for d in (2, 3, 10, 40):
        mean = np.tile(np.ones(d) / np.sqrt(d), (len(ks), 1))
        got = VonMisesFisher(mean=mean, concentration=ks).log_norm()
        np.testing.assert_allclose(got, g[f'log_norm_d{d}'], rtol=1e-12, atol=2e-12)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_single_fits_and_log_pdf_match_reference(dtype):
    from pb_bss_amd.distribution import GaussianTrainer, VonMisesFisherTrainer
    g = load('embed_single_fits')
    y = g['y'].astype(dtype)[None]          # (1, N, E); the fixture data are float32 values
    sal = g['saliency']                     # (K, N)
    v = VonMisesFisherTrainer()._fit(y, saliency=sal, min_concentration=1e-10,
                                     max_concentration=500)
    assert v.mean.shape == g['vmf_mean'].shape and v.concentration.shape == (3,)
    np.testing.assert_allclose(v.mean, g['vmf_mean'], atol=1e-13)
    np.testing.assert_allclose(v.concentration, g['vmf_concentration'], rtol=1e-11)
    m = GaussianTrainer()._fit(y, saliency=sal, covariance_type='spherical')
    np.testing.assert_allclose(m.mean, g['spherical_mean'], atol=1e-13)
    np.testing.assert_allclose(m.covariance, g['spherical_covariance'], rtol=1e-11)
    np.testing.assert_allclose(m.log_pdf(y), g['spherical_log_pdf'], rtol=1e-10, atol=1e-9)
    f = GaussianTrainer()._fit(y, saliency=sal, covariance_type='full')  # FP64 MFMA scatter
    np.testing.assert_allclose(f.mean, g['full_mean'], atol=1e-12)
    np.testing.assert_allclose(f.covariance, g['full_covariance'], atol=1e-12)
    np.testing.assert_allclose(f.log_pdf(y), g['full_log_pdf'], rtol=1e-9, atol=1e-8)
    d = GaussianTrainer()._fit(y, saliency=sal, covariance_type='diagonal')
    np.testing.assert_allclose(d.mean, g['diagonal_mean'], atol=1e-12)
    np.testing.assert_allclose(d.covariance, g['diagonal_covariance'], rtol=1e-10)
    np.testing.assert_allclose(d.log_pdf(y), g['diagonal_log_pdf'], rtol=1e-10, atol=1e-9)
    with pytest.raises(ValueError):
        GaussianTrainer()._fit(y, saliency=sal, covariance_type='tied')


def test_vmfmm_matches_reference_flat_and_independent_axes():
    from pb_bss_amd.distribution import VMFMMTrainer
    g = load('embed_vmfmm_n480_e10_k3')
    model = VMFMMTrainer().fit(g['y'], initialization=g['init'], iterations=int(g['iterations']))
    assert model.vmf.mean.shape == g['mean'].shape
    assert model.vmf.concentration.shape == g['concentration'].shape
    assert model.weight.shape == g['weight'].shape
    np.testing.assert_allclose(model.vmf.mean, g['mean'], atol=1e-10)
    np.testing.assert_allclose(model.vmf.concentration, g['concentration'], rtol=1e-9)
    np.testing.assert_allclose(model.weight, g['weight'], atol=1e-11)
    np.testing.assert_allclose(model.predict(g['y']), g['affiliation'], atol=1e-9)
    np.testing.assert_allclose(model.vmf.log_pdf(g['y'][None]), g['log_pdf'], atol=1e-8)
    g = load('embed_vmfmm_indep_f6_t80_e10_k3')
    model = VMFMMTrainer().fit(g['y'], initialization=g['init'], iterations=int(g['iterations']),
                               saliency=g['saliency'],
                               max_concentration=float(g['max_concentration']))
    assert model.vmf.mean.shape == g['mean'].shape == (6, 3, 10)
    np.testing.assert_allclose(model.vmf.mean, g['mean'], atol=1e-10)
    np.testing.assert_allclose(model.vmf.concentration, g['concentration'], rtol=1e-9)
    np.testing.assert_allclose(model.weight, g['weight'], atol=1e-11)
    np.testing.assert_allclose(model.predict(g['y']), g['affiliation'], atol=1e-9)


@pytest.mark.parametrize('N,E,K', [(1000, 2, 2), (4099, 3, 4), (30000, 40, 3), (2500, 64, 6),
                                   (700, 100, 2), (5, 7, 1), (6000, 40, 8), (3001, 12, 7)])
def test_vmfmm_shapes_against_oracle(N, E, K):
    """odd sample counts, one sample per workgroup slot, E not dividing 256, K = 1..8"""
    from pb_bss_amd.distribution import VMFMMTrainer
    from oracle import embed as oe
    rng = np.random.default_rng(N + E)
    mu = rng.standard_normal((K, E))
    y = (mu[rng.integers(K, size=N)] + 0.7 * rng.standard_normal((N, E))).astype(np.float32)
    init = rng.uniform(size=(K, N))
    init /= init.sum(0)
    sal = rng.uniform(0.1, 1.0, size=N)
    ref = oe.vmfmm_fit(y.astype(np.float64), init, 6, saliency=sal)
    model = VMFMMTrainer().fit(y, initialization=init, iterations=6, saliency=sal)
    np.testing.assert_allclose(model.vmf.mean, ref['mean'], atol=1e-9)
    np.testing.assert_allclose(model.vmf.concentration, ref['concentration'], rtol=1e-8)
    np.testing.assert_allclose(model.weight, ref['weight'], atol=1e-10)
    np.testing.assert_allclose(model.predict(y), oe.vmfmm_predict(ref, y.astype(np.float64)),
                               atol=1e-8)


@pytest.mark.parametrize('B,N,E,K,dtype,uniform', [
    (20, 300, 10, 3, np.float32, False), (257, 800, 12, 3, np.float32, False),
    (33, 257, 7, 2, np.float64, True), (16, 1200, 40, 5, np.float32, False),
    # round-6 kernel (vmf_bin.hip): four waves (<= 4 chunks), 64 accumulators with more
    # mixtures than compute units (four waves), float64 rows at six waves, one chunk only
    (40, 200, 4, 4, np.float32, False), (300, 500, 14, 3, np.float32, True),
    (260, 330, 12, 2, np.float64, False), (17, 50, 16, 2, np.float32, False)])
def test_vmfmm_many_small_mixtures_persistent_kernel(B, N, E, K, dtype, uniform):
    """B >= 16 independent mixtures (one per frequency bin in BASELINE configs[3]) run in the
    persistent one-workgroup-per-mixture kernels (vmf_bin.hip: vmf_bin_em2_kernel for E <= 16
    features and 2 <= K <= 4 classes, embed.hip: vmf_bin_em_kernel otherwise): whole EM loop in
    one launch, model in LDS -- against the oracle's reference loop, with saliency, uniform
    weights, float64 input and the predict of the fitted model (iterations = 0 path)."""
    from pb_bss_amd.distribution import VMFMMTrainer
    from oracle import embed as oe
    rng = np.random.default_rng(B + N)
    mu = rng.standard_normal((B, K, E))
    lab = rng.integers(K, size=(B, N))
    y = (np.take_along_axis(mu, lab[..., None], 1) + 0.6 * rng.standard_normal((B, N, E))).astype(dtype)
    # an initialisation that leans towards the true labels: no class starves (a class that
    # collapses onto a single point has r_bar = 1 +- rounding, its concentration is a coin flip
    # between the two clamps in ANY implementation)
    init = rng.uniform(size=(B, K, N)) + 2.0 * (np.arange(K)[None, :, None] == lab[:, None, :])
    init /= init.sum(1, keepdims=True)
    sal = rng.uniform(0.1, 1.0, size=(B, N))
    kw = dict(weight_constant_axis=-2) if uniform else {}
    y64 = y.astype(np.float64)
    ref = oe.vmfmm_fit(y64, init, 7, saliency=sal, **kw)
    model = VMFMMTrainer().fit(y, initialization=init, iterations=7, saliency=sal, **kw)
    np.testing.assert_allclose(model.vmf.mean, ref['mean'], atol=1e-9)
    np.testing.assert_allclose(model.vmf.concentration, ref['concentration'], rtol=1e-8)
    if not uniform:
        np.testing.assert_allclose(model.weight, ref['weight'], atol=1e-10)
    want = oe.vmfmm_predict(ref, y64)
    np.testing.assert_allclose(model.predict(y), want, atol=1e-8)
    np.testing.assert_allclose(
        VMFMMTrainer().fit_predict(y, initialization=init, iterations=7, saliency=sal, **kw),
        want, atol=1e-8)


def test_vmfmm_torch_in_torch_out_and_uniform_weight():
    import torch
    from pb_bss_amd.distribution import VMFMMTrainer
    from oracle import embed as oe
    rng = np.random.default_rng(5)
    y = rng.standard_normal((3, 400, 8))
    init = rng.uniform(size=(3, 2, 400))
    init /= init.sum(1, keepdims=True)
    yt = torch.as_tensor(y, device='cuda')
    model = VMFMMTrainer().fit(yt, initialization=torch.as_tensor(init, device='cuda'),
                               iterations=4, weight_constant_axis=-2)
    assert model.vmf.mean.is_cuda and model.weight.shape == (2, 1)
    ref = oe.vmfmm_fit(y, init, 4, weight_constant_axis=-2)
    np.testing.assert_allclose(model.vmf.mean.cpu().numpy(), ref['mean'], atol=1e-10)
    aff = model.predict(yt)
    assert aff.is_cuda and aff.shape == (3, 2, 400)
    np.testing.assert_allclose(aff.cpu().numpy(), oe.vmfmm_predict(ref, y), atol=1e-9)


JOINT = [('embed_gcacgmm_spherical', 'gaussian'), ('embed_gcacgmm_weights', 'gaussian'),
         ('embed_gcacgmm_inline_pa', 'gaussian'), ('embed_vmfcacgmm', 'vmf'),
         # covariance_type='full' / 'diagonal' (gcacgmm.py:141): fixtures from the live reference
         ('embed_gcacgmm_full', 'gaussian'), ('embed_gcacgmm_diagonal', 'gaussian'),
         ('embed_gcacgmm_full_weights', 'gaussian'), ('embed_gcacgmm_diagonal_pa', 'gaussian'),
         ('embed_gcacgmm_full_fixed', 'gaussian')]


@pytest.mark.parametrize('name,kind', JOINT)
def test_joint_models_match_reference(name, kind):
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    g = load(name)
    kw = ast.literal_eval(str(g['kwargs']))
    if 'fixed_covariance' in g.files:
        kw['fixed_covariance'] = g['fixed_covariance']
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    model = trainer.fit(g['Y'], g['embedding'], initialization=g['init'],
                        iterations=int(g['iterations']), **kw)
    spec = model.gaussian if kind == 'gaussian' else model.vmf
    assert np.shape(model.weight) == g['weight'].shape
    np.testing.assert_allclose(np.asarray(model.weight), g['weight'], atol=1e-9)
    np.testing.assert_allclose(spec.mean, g['mean'], atol=1e-9)
    if kind == 'gaussian':
        np.testing.assert_allclose(spec.covariance, g['covariance'], rtol=1e-8)
    else:
        np.testing.assert_allclose(spec.concentration, g['concentration'], rtol=1e-8)
    np.testing.assert_allclose(cov(model.cacg.covariance_eigenvectors, model.cacg.covariance_eigenvalues),
                               cov(g['eigvec'], g['eigval']), atol=1e-7)
    aff = model.predict(g['Y'], g['embedding'])
    assert aff.shape == g['affiliation'].shape
    assert np.abs(aff - g['affiliation']).max() < 1e-7
    if kind == 'gaussian':
        from pb_bss_amd.distribution import DiagonalGaussian, Gaussian, SphericalGaussian
        want = {'full': Gaussian, 'diagonal': DiagonalGaussian}.get(kw.get('covariance_type'),
                                                                    SphericalGaussian)
        assert type(model.gaussian) is want


def test_diagonal_gaussian_single_fit_and_log_pdf_match_reference():
    """GaussianTrainer._fit(covariance_type='diagonal') and DiagonalGaussian.log_pdf -- the
    latter exactly as the reference evaluates it (gaussian.py:87-91)."""
    from pb_bss_amd.distribution import DiagonalGaussian, GaussianTrainer
    g = load('embed_single_fits')
    y = g['y'].astype(np.float64)
    m = GaussianTrainer()._fit(y[None], saliency=g['saliency'], covariance_type='diagonal')
    assert isinstance(m, DiagonalGaussian)
    np.testing.assert_allclose(m.mean, g['diagonal_mean'], atol=1e-11)
    np.testing.assert_allclose(m.covariance, g['diagonal_covariance'], rtol=1e-10)
    lp = DiagonalGaussian(mean=g['diagonal_mean'], covariance=g['diagonal_covariance']).log_pdf(y[None])
    np.testing.assert_allclose(lp, g['diagonal_log_pdf'], rtol=1e-10, atol=1e-9)


def test_gcacgmm_full_config5_shape_against_oracle():
    """BASELINE config 5 shape (8 mics, K=3, 40-dim embeddings) with a full-covariance spectral
    model: MFMA scatter / quadratic forms behind the joint EM loop."""
    from pb_bss_amd.distribution import GCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, D, K, E = 24, 300, 8, 3, 40
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=9)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    for ct in ('full', 'diagonal'):
        ref = oe.joint_fit('gaussian', Y128, e64, init, 4, covariance_type=ct)
        model = GCACGMMTrainer().fit(Y, e, initialization=init, iterations=4, covariance_type=ct)
        np.testing.assert_allclose(model.gaussian.mean, ref['mean'], atol=1e-8)
        np.testing.assert_allclose(model.gaussian.covariance, ref['covariance'], rtol=1e-6, atol=1e-9)
        aff = model.predict(Y, e)
        assert np.abs(aff - oe.joint_model_predict(ref, Y128, e64)).max() < 1e-6, ct


@pytest.mark.parametrize('kind,kw', [
    ('gaussian', {}),
    ('gaussian', dict(weight_constant_axis=(-3, -2, -1), spectral_weight=0.5)),
    ('gaussian', dict(weight_constant_axis=(-2,))),
    ('gaussian', dict(weight_constant_axis=(0, 1, 2))),
    ('gaussian', dict(weight_constant_axis=(-3,), inline_permutation_alignment=True)),
    ('vmf', dict(weight_constant_axis=(-3, -1), max_concentration=80.)),
])
def test_joint_models_config5_shape_against_oracle(kind, kw):
    """BASELINE config 5 at reduced F: 8 mics, K=3, 40-dim embeddings, with saliency."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, D, K, E = 24, 300, 8, 3, 40
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=9)
    sal = np.random.default_rng(1).uniform(0.2, 1.0, size=(F, T))
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    masks = trainer.fit_predict(Y, e, initialization=init, iterations=8, saliency=sal, **kw)
    ref = oe.joint_fit(kind, Y.astype(np.complex128), e.astype(np.float64), init, 8,
                       saliency=sal, **kw)
    want = oe.joint_model_predict(ref, Y.astype(np.complex128), e.astype(np.float64))
    assert masks.shape == (F, K, T)
    assert np.abs(masks - want).max() < 1e-6


def test_joint_fixed_covariance_and_errors():
    from pb_bss_amd.distribution import GCACGMMTrainer
    from oracle import embed as oe, synth
    Y, e, init = synth.make_joint(5, 90, 3, 2, 12, seed=2)
    fixed = np.array([0.02, 0.05])
    model = GCACGMMTrainer().fit(Y, e, initialization=init, iterations=4, fixed_covariance=fixed)
    ref = oe.joint_fit('gaussian', Y.astype(np.complex128), e.astype(np.float64), init, 4,
                       fixed_covariance=fixed)
    np.testing.assert_allclose(model.gaussian.covariance, fixed)
    np.testing.assert_allclose(model.gaussian.mean, ref['mean'], atol=1e-9)
    with pytest.raises(ValueError):  # gaussian.py:184 "Unknown covariance type"
        GCACGMMTrainer().fit(Y, e, initialization=init, iterations=2, covariance_type='tied')
    with pytest.raises(AssertionError):  # fixed covariance of the wrong shape (gcacgmm.py:306)
        GCACGMMTrainer().fit(Y, e, initialization=init, iterations=2, covariance_type='full',
                             fixed_covariance=fixed)
    with pytest.raises(AssertionError):
        GCACGMMTrainer().fit(Y, e, initialization=init, num_classes=2)


def test_gmm_matches_reference():
    """GMMTrainer / GMM (spherical) on the device against fixtures of the real reference."""
    from pb_bss_amd.distribution import GMM, GMMTrainer, SphericalGaussian
    g = load('gmm_n450_e12_k3')
    for dtype in (np.float32, np.float64):
        y = g['y'].astype(dtype)
        m = GMMTrainer().fit(y, initialization=g['init'], iterations=int(g['iterations']),
                             covariance_type='spherical')
        assert isinstance(m, GMM) and isinstance(m.gaussian, SphericalGaussian)
        assert m.weight.shape == g['weight'].shape == (3, 1)
        np.testing.assert_allclose(m.gaussian.mean, g['mean'], atol=1e-11)
        np.testing.assert_allclose(m.gaussian.covariance, g['covariance'], rtol=1e-10)
        np.testing.assert_allclose(m.weight, g['weight'], atol=1e-11)
        np.testing.assert_allclose(m.predict(y), g['affiliation'], atol=1e-9)
    g = load('gmm_variants_n450_e12_k3')
    y = g['y'].astype(np.float64)
    it = int(g['iterations'])
    for tag, kw in [('sal', dict(saliency=g['saliency'])),
                    ('uniform', dict(weight_constant_axis=-2)),
                    ('fixed', dict(fixed_covariance=g['fixed']))]:
        m = GMMTrainer().fit(y, initialization=g['init'], iterations=it,
                             covariance_type='spherical', **kw)
        np.testing.assert_allclose(m.gaussian.mean, g[tag + '_mean'], atol=1e-11, err_msg=tag)
        np.testing.assert_allclose(m.gaussian.covariance, g[tag + '_covariance'], rtol=1e-10)
        np.testing.assert_allclose(m.weight, g[tag + '_weight'], atol=1e-11)
        np.testing.assert_allclose(m.predict(y), g[tag + '_affiliation'], atol=1e-9)
    aff = GMMTrainer().fit_predict(y, initialization=g['init'], iterations=3,
                                   covariance_type='spherical')
    np.testing.assert_allclose(aff, g['fit_predict'], atol=1e-9)
    m = GMMTrainer().fit(y, initialization=g['init'], iterations=3, weight_constant_axis=(-2,),
                         covariance_type='spherical')
    assert m.weight.shape == (1, 450) and (m.weight == 1).all()
    np.testing.assert_allclose(m.predict(y), g['fit_predict'], atol=1e-9)
    with pytest.raises(ValueError):
        GMMTrainer().fit(y, initialization=g['init'], iterations=2, covariance_type='round')
    with pytest.raises(AssertionError):
        GMMTrainer().fit(y, initialization=g['init'], num_classes=3, covariance_type='spherical')


@pytest.mark.parametrize('F,N,E,K', [(1, 20000, 40, 3), (7, 333, 5, 2), (3, 1000, 16, 6), (2, 1500, 10, 8)])
def test_gmm_shapes_against_oracle(F, N, E, K):
    """Larger / batched problems (independent leading axis) against the NumPy oracle; torch in,
    torch out; num_classes initialisation draws from the global NumPy RNG like the reference."""
    import torch
    from oracle import embed as oe
    from pb_bss_amd.distribution import GMMTrainer
    rng = np.random.default_rng(F * 100 + K)
    centers = rng.normal(size=(F, K, E)) * 2
    lab = rng.integers(K, size=(F, N))
    y = (np.take_along_axis(centers, lab[..., None], 1) +
         rng.normal(size=(F, N, E)) * rng.uniform(0.3, 1.0, size=(F, 1, 1))).astype(np.float32)
    init = rng.uniform(size=(F, K, N))
    init /= init.sum(-2, keepdims=True)
    sal = rng.uniform(0.1, 1.0, size=(F, N))
    y64 = y.astype(np.float64)
    o = oe.gmm_fit(y64, init, 6, saliency=sal)
    m = GMMTrainer().fit(y, initialization=init, iterations=6, saliency=sal,
                         covariance_type='spherical')
    np.testing.assert_allclose(m.gaussian.mean, o['mean'], atol=1e-9)
    np.testing.assert_allclose(m.gaussian.covariance, o['covariance'], rtol=1e-9)
    np.testing.assert_allclose(m.weight, o['weight'], atol=1e-10)
    np.testing.assert_allclose(m.predict(y), oe.gmm_predict(o, y64), atol=1e-7)
    mt = GMMTrainer().fit(torch.as_tensor(y).cuda(), initialization=torch.as_tensor(init).cuda(),
                          iterations=6, saliency=torch.as_tensor(sal).cuda(),
                          covariance_type='spherical')
    assert mt.gaussian.mean.is_cuda and mt.weight.shape == (F, K, 1)
    np.testing.assert_allclose(mt.gaussian.mean.cpu().numpy(), m.gaussian.mean, atol=0)
    np.random.seed(5)
    m1 = GMMTrainer().fit(y, num_classes=K, iterations=3, covariance_type='spherical')
    np.random.seed(5)
    i1 = np.random.uniform(size=(F, K, N))
    i1 /= np.einsum('...kn->...n', i1)[..., None, :]
    o1 = oe.gmm_fit(y64, i1, 3)
    np.testing.assert_allclose(m1.gaussian.mean, o1['mean'], atol=1e-9)


@pytest.mark.parametrize('N,E,K', [(20000, 40, 3), (333, 5, 2), (1000, 16, 6), (4100, 63, 1),
                                   (70, 17, 4), (600, 50, 3), (900, 10, 40), (500, 47, 2)])
def test_gaussian_full_covariance_against_oracle(N, E, K):
    """GaussianTrainer(covariance_type='full') / Gaussian.log_pdf on the matrix pipe against the
    NumPy oracle (which is pinned to the reference by embed_single_fits): every tile count
    (E + 1 <= 16, 32, 48, 64; E + 1 = 48 exactly), ragged sample counts, float32 / float64
    input, no saliency; class counts beyond one LDS pass of whitening matrices (E = 50: one
    class per pass; K = 40: two passes, one sample group per wave) and every per-thread
    position count of the factorisation (E = 5 ... 63)."""
    from oracle import embed as oe
    from pb_bss_amd.distribution import Gaussian, GaussianTrainer
    rng = np.random.default_rng(N + E)
    A = 0.6 * np.eye(E) + 0.4 * rng.normal(size=(E, E)) / np.sqrt(E)  # cond(cov) ~ 1e2
    y32 = (rng.normal(size=(N, E)) @ A + rng.normal(size=E) * 3).astype(np.float32)
    w = rng.uniform(size=(K, N)) ** 2
    for y in (y32, y32.astype(np.float64)):
        y64 = y.astype(np.float64)
        m = GaussianTrainer().fit(y[None], saliency=w, covariance_type='full')
        assert isinstance(m, Gaussian) and m.covariance.shape == (K, E, E)
        om, oc = oe.gaussian_fit(y64[None], w, 'full')
        scale = np.abs(oc).max()
        np.testing.assert_allclose(m.mean, om, atol=1e-11)
        np.testing.assert_allclose(m.covariance, oc, atol=1e-11 * scale)
        assert (m.covariance == np.swapaxes(m.covariance, -1, -2)).all()
        lp = m.log_pdf(y[None])
        ref = oe.gaussian_log_pdf(y64[None], om, oc, 'full')
        np.testing.assert_allclose(lp, ref, rtol=1e-9, atol=1e-7)  # cond(cov) * eps * |lp|
        np.testing.assert_allclose(m.log_det_precision_cholesky,
                                   np.sum(np.log(np.diagonal(np.swapaxes(np.linalg.inv(
                                       np.linalg.cholesky(oc)), -1, -2), axis1=-2, axis2=-1)), -1),
                                   rtol=1e-10)
    m1 = GaussianTrainer().fit(y32, covariance_type='full')          # saliency None
    o1 = oe.gaussian_fit(y32.astype(np.float64), None, 'full')
    np.testing.assert_allclose(m1.covariance, o1[1], atol=1e-11 * np.abs(o1[1]).max())
    with pytest.raises(ValueError, match='ill-defined'):  # sklearn's error in the reference
        Gaussian(mean=np.zeros(E), covariance=-np.eye(E)).log_pdf(y32)


def test_gmm_full_covariance_matches_reference_and_oracle():
    """GMMTrainer with its default covariance_type='full' (FP64 matrix-pipe kernels) against the
    fixture of the real reference, and a larger batched problem against the NumPy oracle."""
    from oracle import embed as oe
    from pb_bss_amd.distribution import GMM, GMMTrainer, Gaussian
    g = load('gmm_full_n450_e12_k3')
    for dtype in (np.float32, np.float64):
        y = g['y'].astype(dtype)
        m = GMMTrainer().fit(y, initialization=g['init'], iterations=int(g['iterations']))
        assert isinstance(m, GMM) and isinstance(m.gaussian, Gaussian)
        assert m.gaussian.covariance.shape == (3, 12, 12) and m.weight.shape == (3, 1)
        np.testing.assert_allclose(m.gaussian.mean, g['mean'], atol=1e-9)
        np.testing.assert_allclose(m.gaussian.covariance, g['covariance'], atol=1e-9)
        np.testing.assert_allclose(m.weight, g['weight'], atol=1e-9)
        np.testing.assert_allclose(m.predict(y), g['affiliation'], atol=1e-7)
    y = g['y'].astype(np.float64)
    m = GMMTrainer().fit(y, initialization=g['init'], iterations=4, saliency=g['saliency'],
                         fixed_covariance=g['fixed'])
    np.testing.assert_allclose(m.gaussian.mean, g['fixed_mean'], atol=1e-9)
    np.testing.assert_allclose(m.weight, g['fixed_weight'], atol=1e-9)
    assert (m.gaussian.covariance == g['fixed']).all()
    np.testing.assert_allclose(m.predict(y), g['fixed_affiliation'], atol=1e-7)
    np.testing.assert_allclose(
        GMMTrainer().fit_predict(y, initialization=g['init'], iterations=3), g['fit_predict'],
        atol=1e-7)
    # batch of mixtures, config-5-sized embedding dimension
    rng = np.random.default_rng(8)
    F, N, E, K = 3, 6000, 40, 3
    centers = rng.normal(size=(F, K, E)) * 1.5
    lab = rng.integers(K, size=(F, N))
    yb = (np.take_along_axis(centers, lab[..., None], 1) + rng.normal(size=(F, N, E))).astype(np.float32)
    init = rng.uniform(size=(F, K, N))
    init /= init.sum(-2, keepdims=True)
    yb64 = yb.astype(np.float64)
    o = oe.gmm_fit(yb64, init, 5, covariance_type='full')
    mb = GMMTrainer().fit(yb, initialization=init, iterations=5)
    np.testing.assert_allclose(mb.gaussian.mean, o['mean'], atol=1e-8)
    np.testing.assert_allclose(mb.gaussian.covariance, o['covariance'], atol=1e-8)
    np.testing.assert_allclose(mb.weight, o['weight'], atol=1e-9)
    np.testing.assert_allclose(mb.predict(yb), oe.gmm_predict(o, yb64, 'full'), atol=1e-6)
    # a class that collapses onto a single point has a singular covariance: ValueError like
    # sklearn's precision Cholesky in the reference
    bad = np.zeros((2, 50))
    bad[0, 0] = 1.0
    bad[1, 1:] = 1.0
    with pytest.raises(ValueError, match='ill-defined'):
        GMMTrainer().fit(yb[0, :50], initialization=bad, iterations=2)


@pytest.mark.parametrize('kind,kw', [('gaussian', {}), ('vmf', {}),
                                     ('gaussian', dict(weight_constant_axis=(-3,))),
                                     ('vmf', dict(weight_constant_axis=(-3, -1)))])
def test_joint_remainder_bins_as_member_workgroups(kind, kw):
    """B = CUs + r bins: the r remainder bins of every one-iteration joint launch run as member
    workgroups on frame windows (run_joint_member: partial sums through L2, the last arriver
    factors).  Same masks and model as the plain launch (split handling off) and as the oracle;
    258 bins = 256 CUs + 2 on an MI355X, 300 frames = four full windows and one of 44; per-class
    weights per bin, per frame (shared over the bins) and per utterance."""
    from pb_bss_amd import engine
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, D, K, E = 258, 300, 4, 3, 12
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=21)
    sal = np.random.default_rng(2).uniform(0.2, 1.0, size=(F, T))
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    out = {}
    try:
        for split in (True, False):
            engine.set_split_tail(split)
            model = trainer.fit(Y, e, initialization=init, iterations=5, saliency=sal, **kw)
            out[split] = (model, model.predict(Y, e))
    finally:
        engine.set_split_tail(True)
    on, off = out[True], out[False]
    assert np.abs(on[1] - off[1]).max() < 1e-9
    np.testing.assert_allclose(on[0].cacg.covariance_eigenvalues, off[0].cacg.covariance_eigenvalues,
                               rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(on[0].weight, off[0].weight, atol=1e-12)
    ref = oe.joint_fit(kind, Y.astype(np.complex128), e.astype(np.float64), init, 5, saliency=sal,
                       **kw)
    want = oe.joint_model_predict(ref, Y.astype(np.complex128), e.astype(np.float64))
    assert np.abs(on[1] - want).max() < 1e-6
    assert np.abs(on[1][-2:] - want[-2:]).max() < 1e-6   # the two member-handled bins


@pytest.mark.parametrize('kind', ['gaussian', 'vmf'])
def test_joint_models_config5_full_size(kind):
    """BASELINE configs[4] at its full size: F = 513, T = 500, 8 mics, K = 3, 40-dim embeddings
    (256 500 points in ONE spectral mixture; the 513th bin as member workgroups), 4 EM
    iterations against the oracle's reference loop."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, D, K, E = 513, 500, 8, 3, 40
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=3)
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    kw = {} if kind == 'gaussian' else dict(max_concentration=80.)
    masks = trainer.fit_predict(Y, e, initialization=init, iterations=4, **kw)
    ref = oe.joint_fit(kind, Y.astype(np.complex128), e.astype(np.float64), init, 4, **kw)
    want = oe.joint_model_predict(ref, Y.astype(np.complex128), e.astype(np.float64))
    assert masks.shape == (F, K, T)
    assert np.abs(masks - want).max() < 1e-6
    assert np.abs(masks[512] - want[512]).max() < 1e-6


@pytest.mark.parametrize('kind', ['gaussian', 'vmf'])
def test_joint_models_long_utterance_runs_on_the_streaming_kernels(kind):
    """An utterance too long for the LDS-resident joint kernels (T = 3 000 frames at D = 4, K = 2:
    168 KB of frame arrays) is served by the size-generic kernels instead of being refused -- the
    reference has no length limit (gcacgmm.py:121-225)."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, D, K, E = 3, 3000, 4, 2, 12
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=8)
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    kw = {} if kind == 'gaussian' else dict(max_concentration=80.)
    masks = trainer.fit_predict(Y, e, initialization=init, iterations=4, **kw)
    ref = oe.joint_fit(kind, Y.astype(np.complex128), e.astype(np.float64), init, 4, **kw)
    want = oe.joint_model_predict(ref, Y.astype(np.complex128), e.astype(np.float64))
    assert masks.shape == (F, K, T)
    assert np.abs(masks - want).max() < 1e-6


def _joint_trajectory(kind, F, T, iterations, seed, **kw):
    """Device fit + predict and the oracle's, same inputs -> (model, masks, oracle model, oracle
    masks) of a config-5-shaped problem (8 mics, K = 3, 40-dim embeddings) with F bins."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    D, K, E = 8, 3, 40
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=seed)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    model = trainer.fit(Y, e, initialization=init, iterations=iterations, **kw)
    masks = model.predict(Y, e)
    ref = oe.joint_fit(kind, Y128, e64, init, iterations, **kw)
    want = oe.joint_model_predict(ref, Y128, e64)
    return model, masks, ref, want


def _assert_joint_model_close(model, ref, kind, tol):
    """Spectral model (mean + covariance / concentration), cACG covariance V diag(lambda) V^H
    (eigenvector phases are arbitrary, the product is not) and class weights."""
    np.testing.assert_allclose(np.squeeze(model.weight), np.squeeze(ref['weight']), atol=tol)
    got = cov(model.cacg.covariance_eigenvectors, model.cacg.covariance_eigenvalues)
    np.testing.assert_allclose(got, cov(ref['eigvec'], ref['eigval']), atol=tol)
    if kind == 'gaussian':
        np.testing.assert_allclose(model.gaussian.mean, ref['mean'], atol=tol)
        np.testing.assert_allclose(model.gaussian.covariance, ref['covariance'], rtol=tol, atol=tol)
    else:
        np.testing.assert_allclose(model.vmf.mean, ref['mean'], atol=tol)
        np.testing.assert_allclose(model.vmf.concentration, ref['concentration'], rtol=tol)


@pytest.mark.parametrize('kind', ['gaussian', 'vmf'])
def test_joint_models_100_iterations_against_oracle(kind):
    """The trajectory bench.py times (configs[4]: 100 EM iterations), pinned end to end: every
    iteration but the last goes through the Gauss-Jordan fast path with the conditioning veto
    (gcacgmm.py:121-225; complex_angular_central_gaussian.py:112-126 is evaluated exactly only
    where the floor could bind), so a drift of that path would show here and nowhere in the
    4 / 5 / 8-iteration cases above.  F = 65 keeps the oracle at a few seconds; the spectral
    mixture couples the bins, so this is a self-contained problem, not a slice of the big one."""
    kw = {} if kind == 'gaussian' else dict(max_concentration=80.)
    model, masks, ref, want = _joint_trajectory(kind, 65, 500, 100, seed=3, **kw)
    assert np.abs(masks - want).max() < 1e-6
    _assert_joint_model_close(model, ref, kind, 1e-7)


def test_gcacgmm_full_covariance_50_iterations_against_oracle():
    """covariance_type='full' (gaussian.py:152-193 behind gcacgmm.py:267-333): 50 iterations of the
    FP64-MFMA scatter / Cholesky path inside the joint loop."""
    model, masks, ref, want = _joint_trajectory('gaussian', 17, 500, 50, seed=4,
                                                covariance_type='full')
    assert np.abs(masks - want).max() < 1e-6
    _assert_joint_model_close(model, ref, 'gaussian', 1e-7)


@pytest.mark.parametrize('kind', ['gaussian', 'vmf'])
def test_joint_inline_permutation_alignment_50_iterations_against_oracle(kind):
    """inline_permutation_alignment=True (mixture_model_utils.py:58-130) over 50 iterations: the
    per-bin permutation search must take the same decision as the reference in every bin of every
    iteration, otherwise the trajectories part for good."""
    kw = dict(weight_constant_axis=(-3,), inline_permutation_alignment=True)
    if kind == 'vmf':
        kw['max_concentration'] = 80.
    model, masks, ref, want = _joint_trajectory(kind, 33, 500, 50, seed=5, **kw)
    assert np.abs(masks - want).max() < 1e-6
    _assert_joint_model_close(model, ref, kind, 1e-7)


@pytest.mark.parametrize('E,K', [(52, 2), (44, 4), (8, 4), (64, 3), (30, 3)])
def test_joint_sweep_wavefront_kernel_dimension_range(E, K):
    """The sweep of the rotated joint loop has a wavefront-per-64-rows kernel since round 6
    (embed.hip: joint_sweep_wave_kernel; float32 embeddings, E % 4 == 0, K = 2 ... 4, LDS <= 64 KB)
    with 10 or 16 sixteen-byte pieces per lane: dimensions on both sides of the 40 / 64 limits and
    the ones it refuses (E = 64 at K = 3: LDS; E = 30: not whole pieces) must all agree with the
    oracle -- whichever kernel served them.  Ragged last chunk (F T = 3 x 341 rows), saliency."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, T, D = 3, 341, 4
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=E + K)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    sal = np.random.default_rng(E).uniform(0.1, 1.0, size=(F, T))
    for kind, trainer, kw in (('gaussian', GCACGMMTrainer(), {}),
                              ('vmf', VMFCACGMMTrainer(), dict(max_concentration=80.))):
        got = trainer.fit_predict(Y, e, initialization=init, iterations=4, saliency=sal, **kw)
        ref = oe.joint_fit(kind, Y128, e64, init, 4, saliency=sal, **kw)
        assert np.abs(got - oe.joint_model_predict(ref, Y128, e64)).max() < 1e-6, (E, K, kind)


@pytest.mark.parametrize('T', [64, 65, 130, 257])
def test_joint_models_frame_counts_around_the_chunk_size(T):
    """The spatial kernels of the joint models share the 64-frame chunk layout of the LDS frame
    arrays (round 5): frame counts on both sides of the chunk / pass boundaries, the rotated
    two-kernel loop (spherical Gaussian, vMF) and the three-kernel path (diagonal Gaussian)."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from oracle import embed as oe, synth
    F, D, K, E = 5, 6, 3, 12
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=T)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    sal = np.random.default_rng(T).uniform(0.1, 1.0, size=(F, T))
    for kind, trainer, kw in (('gaussian', GCACGMMTrainer(), {}),
                              ('gaussian', GCACGMMTrainer(), dict(covariance_type='diagonal')),
                              ('vmf', VMFCACGMMTrainer(), dict(max_concentration=80.))):
        got = trainer.fit_predict(Y, e, initialization=init, iterations=5, saliency=sal, **kw)
        ref = oe.joint_fit(kind, Y128, e64, init, 5, saliency=sal, **kw)
        assert np.abs(got - oe.joint_model_predict(ref, Y128, e64)).max() < 1e-6, (T, kind, kw)
