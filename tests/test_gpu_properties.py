"""GPU: size-independent properties of the hot path at BASELINE's FULL sizes
(F=513, T=500, D=8, K=3; batches of utterances), where the NumPy oracle is too slow to
be the checker: problem independence (bitwise), class-permutation equivariance, scale
invariance, complex64 == complex128 on identical values, normalisation of the
posteriors, eigen-equations of the beamformers, linearity of the filter application,
idempotence of the permutation alignment.  Everything goes through the public API / C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F, T, D, K = 513, 500, 8, 3


@pytest.fixture(scope='module')
def full():
    from oracle import synth
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(F, T, D, K, seed=0)
    model = CACGMMTrainer().fit(Y, initialization=init, iterations=100)
    masks = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=100)
    # fit_predict returns the kernel's own final E-step; model.predict rebuilds B^-1 from the
    # returned (V, lambda) in a second launch: same numbers up to rounding
    np.testing.assert_allclose(model.predict(Y), masks, atol=1e-12)
    return Y, init, model, masks


def test_posteriors_are_a_partition_of_unity(full):
    Y, init, model, masks = full
    assert masks.shape == (F, K, T)
    assert np.isfinite(masks).all() and masks.min() >= 0.0 and masks.max() <= 1.0
    np.testing.assert_allclose(masks.sum(axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(model.weight.sum(axis=-2), 1.0, atol=1e-12)
    lam = model.cacg.covariance_eigenvalues
    assert (np.diff(lam, axis=-1) >= 0).all() and np.allclose(lam[..., -1], 1.0)   # eigh order, 'eigenvalue' norm
    v = model.cacg.covariance_eigenvectors
    eye = np.einsum('...dk,...dl->...kl', v.conj(), v)
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(D), eye.shape), atol=1e-12)


def test_frequency_bins_are_independent_problems_bitwise(full):
    """A sub-range of bins fitted alone gives bit-identical masks (what makes the
    frequency sharding over GPUs exact); bin 512 takes the split-group path in the full fit."""
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init, _, masks = full
    for lo, hi in ((100, 164), (0, 1), (448, 512)):
        sub = CACGMMTrainer().fit_predict(Y[lo:hi], initialization=init[lo:hi], iterations=100)
        assert np.array_equal(sub, masks[lo:hi]), (lo, hi)
    tail = CACGMMTrainer().fit_predict(Y[512:], initialization=init[512:], iterations=100)
    np.testing.assert_allclose(tail, masks[512:], atol=1e-9)   # split groups sum in another order


def test_batch_of_utterances_equals_single_fits():
    """BASELINE config 3 layout: (B, F, T, D) in one call == per-utterance calls."""
    from oracle import synth
    from pb_bss_amd.distribution import CACGMMTrainer
    pairs = [synth.make_stft(129, 300, D, K, seed=s) for s in (3, 4, 5)]
    Yb = np.stack([p[0] for p in pairs])
    ib = np.stack([p[1] for p in pairs])
    together = CACGMMTrainer().fit_predict(Yb, initialization=ib, iterations=30)
    assert together.shape == (3, 129, K, 300)
    for u in range(3):
        alone = CACGMMTrainer().fit_predict(Yb[u], initialization=ib[u], iterations=30)
        assert np.array_equal(alone, together[u])


def test_class_permutation_equivariance(full):
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init, _, masks = full
    perm = [2, 0, 1]
    got = CACGMMTrainer().fit_predict(Y[:64], initialization=init[:64][:, perm], iterations=100)
    np.testing.assert_allclose(got, masks[:64][:, perm], atol=1e-8)


def test_scale_and_precision_invariance(full):
    """The model only sees y / |y|: powers of two are exact, complex128 input with the
    same values runs the float64-staging kernel and must agree with the float32-staging one."""
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init, _, masks = full
    sl = slice(200, 264)
    scaled = CACGMMTrainer().fit_predict(Y[sl] * np.float32(0.125), initialization=init[sl],
                                         iterations=100)
    assert np.array_equal(scaled, masks[sl])
    wide = CACGMMTrainer().fit_predict(Y[sl].astype(np.complex128), initialization=init[sl],
                                       iterations=100)
    np.testing.assert_allclose(wide, masks[sl], atol=1e-9)
    rot = CACGMMTrainer().fit_predict((Y[sl].astype(np.complex128) * np.exp(0.7j)),
                                      initialization=init[sl], iterations=100)
    np.testing.assert_allclose(rot, masks[sl], atol=1e-7)


def test_beamformer_equations_at_full_size(full):
    from pb_bss_amd import extraction as ex
    Y, _, _, masks = full
    X = Y.transpose(0, 2, 1)
    psd = ex.get_power_spectral_density_matrix(X, masks)
    assert psd.shape == (F, K, D, D)
    np.testing.assert_allclose(psd, psd.conj().swapaxes(-1, -2), atol=1e-12)
    target, noise = psd[:, 0], psd[:, 1] + psd[:, 2]
    w = ex.get_gev_vector(target, noise)
    # generalised eigenpair: T w = lambda N w with w^H N w = 1, lambda = w^H T w the largest
    lam = np.einsum('fd,fde,fe->f', w.conj(), target, w).real
    np.testing.assert_allclose(np.einsum('fd,fde,fe->f', w.conj(), noise, w).real, 1.0, atol=1e-9)
    res = np.einsum('fde,fe->fd', target, w) - lam[:, None] * np.einsum('fde,fe->fd', noise, w)
    scale = np.linalg.norm(np.einsum('fde,fe->fd', target, w), axis=-1)
    assert (np.linalg.norm(res, axis=-1) <= 1e-8 * scale).all()
    # MVDR with an ATF: distortionless response w^H h = 1
    h = ex.get_pca_vector(target)
    wm = ex.get_mvdr_vector(h, noise)
    np.testing.assert_allclose(np.einsum('fd,fd->f', wm.conj(), h), 1.0, atol=1e-9)
    # LCMV with both speakers as constraints: pass the first, null the second
    atf = np.stack([ex.get_pca_vector(psd[:, 0]), ex.get_pca_vector(psd[:, 1])])
    wl = ex.get_lcmv_vector(atf, [1, 0], psd[:, 2])
    resp = np.einsum('kfd,fd->fk', atf.conj(), wl)
    np.testing.assert_allclose(resp, np.broadcast_to([1.0, 0.0], resp.shape), atol=1e-7)
    # filter application is linear in the observation, BAN only rescales
    wb = ex.blind_analytic_normalization(w, noise)
    cs = np.abs(np.einsum('fd,fd->f', wb.conj(), w)) / (np.linalg.norm(wb, axis=-1) * np.linalg.norm(w, axis=-1))
    np.testing.assert_allclose(cs, 1.0, atol=1e-12)
    x1, x2 = X[..., :250], X[..., 250:]
    whole = ex.apply_beamforming_vector(wb, X)
    np.testing.assert_allclose(np.concatenate([ex.apply_beamforming_vector(wb, x1),
                                               ex.apply_beamforming_vector(wb, x2)], axis=-1),
                               whole, atol=1e-12)
    np.testing.assert_allclose(ex.apply_beamforming_vector(wb, 2 * X), 2 * whole, rtol=1e-12)


def test_permutation_alignment_is_a_permutation_with_an_inverse(full):
    """(the synthetic sources are independent across bins, so the alignment itself has no
    ground truth here: identical mappings to the reference are checked in
    test_gpu_permutation_alignment.py; this is the structure at full size)"""
    from oracle import permutation_alignment as opa
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    _, _, _, masks = full
    pa = DHTVPermutationAlignment.from_stft_size(1024)
    m = np.ascontiguousarray(masks.transpose(1, 0, 2))    # (K, F, T)
    mapping = pa.calculate_mapping(m)
    assert mapping.shape == (K, F)
    assert (np.sort(mapping, axis=0) == np.arange(K)[:, None]).all()
    aligned = pa.apply_mapping(m, mapping)
    np.testing.assert_allclose(aligned.sum(axis=0), 1.0, atol=1e-12)
    inverse = np.argsort(mapping, axis=0)
    assert np.array_equal(pa.apply_mapping(aligned, inverse), m)
    assert np.array_equal(mapping, opa.dhtv_calculate_mapping(m, pa.alignment_plan))


def test_embedding_mixture_properties_at_config5_size():
    """vMF mixture on N = F*T points, E = 40: posteriors sum to one, means are unit vectors,
    concentration stays inside the clip range, sample order is irrelevant up to rounding."""
    from oracle import synth
    from pb_bss_amd.distribution import VMFMMTrainer
    _, e, init = synth.make_joint(F, T, 2, K, 40, seed=1)
    y = e.reshape(-1, 40)
    g0 = init.transpose(1, 0, 2).reshape(K, -1)
    model = VMFMMTrainer().fit(y, initialization=g0, iterations=20)
    aff = model.predict(y)
    assert aff.shape == (K, F * T)
    np.testing.assert_allclose(aff.sum(axis=0), 1.0, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(model.vmf.mean, axis=-1), 1.0, atol=1e-12)
    assert (model.vmf.concentration >= 1e-10).all() and (model.vmf.concentration <= 500).all()
    p = np.random.default_rng(2).permutation(F * T)
    shuffled = VMFMMTrainer().fit(y[p], initialization=g0[:, p], iterations=20)
    np.testing.assert_allclose(shuffled.vmf.mean, model.vmf.mean, atol=1e-9)
    np.testing.assert_allclose(shuffled.vmf.concentration, model.vmf.concentration, rtol=1e-9)
