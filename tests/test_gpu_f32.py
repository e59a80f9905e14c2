"""GPU: the packed-FP32 "reference precision" instantiation of the EM kernel
(csrc/cacgmm_em32.hpp, pbbss_em_opts.precision = PBBSS_PRECISION_F32).

Single precision cannot be validated by trajectory max-abs (the reference's
own float32 path drifts 2e-2 from its float64 path over 100 iterations; EM trajectories are
chaotic before convergence).  As the survey prescribes, the kernel is pinned
  * per step, from identical state, against the reference's float32 path (fixtures generated from
    the unmodified reference with complex64 input + ndarray initialisation) and against the
    float64 oracle, within the error the reference itself shows between its two precisions;
  * statistically, with the tolerances of the reference's own test
    (tests/test_distribution/test_cacgmm.py:47-49: covariance atol 0.1, weight atol 0.15);
  * structurally: member workgroups of the remainder bin against the plain launch, model
    initialisation, saliency, every compiled D and K, and the refusals.
"""
import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'cacgmm_single_precision_path.npz')


def _fit32(Y, init, iterations, **kw):
    from pb_bss_amd import _lib, engine
    B, T, D = Y.shape
    K = init.shape[1]
    r = engine.em_fit(_lib.to_device(Y), K, gamma0=_lib.to_device(init), iterations=iterations,
                      final_predict=True, precision='f32', **kw)
    return {k: (None if v is None else _lib.to_host(v)) for k, v in r.items()}


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
@pytest.mark.parametrize('iters', [1, 2])
def test_per_step_against_the_references_float32_path(tag, iters):
    """Device (packed FP32) vs the reference's own complex64 / float32 run, and vs its float64
    run on the same values: the device must be at least as close to float64 as the reference's
    float32 path is (its class sums and factorisation are float64), with a small absolute floor."""
    g = np.load(GOLD)
    Y, init = g[f'{tag}_Y'], g[f'{tag}_init']
    r = _fit32(Y, init, iters)
    aff32 = g[f'{tag}_aff32_it{iters}'].astype(np.float64)
    aff64 = g[f'{tag}_aff64_it{iters}']
    ref_gap = np.abs(aff32 - aff64).max()          # what single precision costs the reference
    dev_gap64 = np.abs(r['affiliation'] - aff64).max()
    dev_gap32 = np.abs(r['affiliation'] - aff32).max()
    assert dev_gap64 < max(2.0 * ref_gap, 2e-5), (dev_gap64, ref_gap)
    assert dev_gap32 < 1e-4, dev_gap32             # per-step tolerance
    cov = np.einsum('...ij,...j,...kj->...ik', r['eigvec'], r['eigval'], r['eigvec'].conj())
    cov32 = g[f'{tag}_cov32_it{iters}'].astype(np.complex128)
    assert np.abs(cov - cov32).max() < 1e-4 * max(1.0, np.abs(cov32).max())
    assert np.abs(r['weight'] - g[f'{tag}_weight32_it{iters}'][..., 0]).max() < 1e-5
    assert int(r['status'].max()) & 3 == 0


@pytest.mark.parametrize('tag', ['k5', 'k6'])
@pytest.mark.parametrize('iters', [1, 2])
def test_five_and_six_classes_against_the_references_float32_path(tag, iters):
    """K = 5, 6 in the packed-FP32 kernel (refused until round 4; two weight vectors per frame,
    two waves per SIMD): per step against the reference's own float32 and float64 runs."""
    g = np.load(os.path.join(os.path.dirname(GOLD), 'cacgmm_single_precision_k56.npz'))
    Y, init = g[f'{tag}_Y'], g[f'{tag}_init']
    r = _fit32(Y, init, iters)
    aff32 = g[f'{tag}_aff32_it{iters}'].astype(np.float64)
    aff64 = g[f'{tag}_aff64_it{iters}']
    ref_gap = np.abs(aff32 - aff64).max()
    assert np.abs(r['affiliation'] - aff64).max() < max(2.0 * ref_gap, 2e-5)
    assert np.abs(r['affiliation'] - aff32).max() < 1e-4
    assert np.abs(r['weight'] - g[f'{tag}_weight32_it{iters}'][..., 0]).max() < 1e-5
    assert int(r['status'].max()) & 3 == 0


@pytest.mark.parametrize('T', [300, 500])   # 5 members < K: every member factors; 8 >= K: distributed
def test_remainder_bin_members_with_six_classes(T):
    from oracle import cacgmm as oc, synth
    from pb_bss_amd import engine
    F, D, K = 257, 4, 6
    Y, init = synth.make_stft(F, T, D, K, seed=T)
    r_split = _fit32(Y, init, 3)
    engine.set_split_tail(False)
    try:
        r_plain = _fit32(Y, init, 3)
    finally:
        engine.set_split_tail(True)
    assert engine.split_error() == 0
    assert np.abs(r_split['affiliation'][:256] - r_plain['affiliation'][:256]).max() == 0.0
    assert np.abs(r_split['affiliation'][256] - r_plain['affiliation'][256]).max() < 2e-4
    sel = [0, 255, 256]
    Y128 = Y[sel].astype(np.complex128)
    ref = oc.em_predict(oc.em_fit(Y128, init[sel], iterations=3), Y128)
    assert np.abs(r_split['affiliation'][sel] - ref).max() < 1e-3
    assert int(r_split['status'].max()) & 3 == 0


@pytest.mark.parametrize('iters', [1, 2])
def test_source_activity_mask_against_the_references_float32_path(iters):
    """source_activity_mask in the packed-FP32 kernel (refused until round 4): per step against
    the reference's own complex64 / float32 run with the same mask -- some frames with every class
    switched off -- and against its float64 run; the float64 kernel on the same input as well."""
    from pb_bss_amd import _lib
    g = np.load(os.path.join(os.path.dirname(GOLD), 'cacgmm_single_precision_mask.npz'))
    Y, init, mask = g['Y'], g['init'], g['mask']
    act = _lib.to_device(mask.astype(np.uint8))
    r = _fit32(Y, init, iters, activity=act)
    aff32 = g[f'aff32_it{iters}'].astype(np.float64)
    aff64 = g[f'aff64_it{iters}']
    ref_gap = np.abs(aff32 - aff64).max()
    assert np.abs(r['affiliation'] - aff64).max() < max(2.0 * ref_gap, 2e-5)
    assert np.abs(r['affiliation'] - aff32).max() < 1e-4
    assert np.abs(r['weight'] - g[f'weight32_it{iters}'][..., 0]).max() < 1e-5
    assert int(r['status'].max()) & 3 == 0
    from pb_bss_amd import engine
    r64 = engine.em_fit(_lib.to_device(Y), init.shape[1], gamma0=_lib.to_device(init),
                        iterations=iters, final_predict=True, activity=act)
    assert np.abs(_lib.to_host(r64['affiliation']) - aff64).max() < 1e-9
    # the trainer takes the packed kernel for this call now (reference arithmetic)
    import pb_bss_amd
    from pb_bss_amd.distribution import CACGMMTrainer
    with pb_bss_amd.arithmetic('reference'):
        m = CACGMMTrainer().fit(Y, initialization=init, iterations=iters, source_activity_mask=mask)
    assert np.abs(m.predict(Y) - aff32).max() < 1e-4
    # ... which kernel served it is visible in the model: bit-identical to the packed kernel's
    # (rounded to the reference's float32 result dtype), not to the float64 kernel's
    cov = lambda vec, val: np.einsum('...ij,...j,...kj->...ik', vec, val, vec.conj())  # noqa: E731
    got = cov(m.cacg.covariance_eigenvectors, m.cacg.covariance_eigenvalues)
    c32 = cov(r['eigvec'].astype(got.dtype), r['eigval'].astype(m.cacg.covariance_eigenvalues.dtype))
    c64 = cov(_lib.to_host(r64['eigvec']).astype(got.dtype),
              _lib.to_host(r64['eigval']).astype(m.cacg.covariance_eigenvalues.dtype))
    assert np.array_equal(got, c32)
    assert not np.array_equal(got, c64)


@pytest.mark.parametrize('D,K,T', [(2, 2, 90), (3, 1, 130), (4, 3, 257), (5, 4, 300), (6, 2, 64),
                                   (7, 3, 511), (8, 4, 500), (8, 3, 256), (2, 5, 130), (3, 6, 257),
                                   (5, 5, 300), (7, 6, 200), (8, 5, 500), (8, 6, 500)])
def test_every_compiled_size_against_the_float64_oracle(D, K, T):
    from oracle import cacgmm as oc, synth
    F = 7
    Y, init = synth.make_stft(F, T, D, K, seed=D * 10 + K)
    sal = np.random.default_rng(T).uniform(0.3, 1.0, size=(F, T))
    from pb_bss_amd import _lib
    r = _fit32(Y, init, 2, saliency=_lib.to_device(sal))
    Y128 = Y.astype(np.complex128)
    ref = oc.em_predict(oc.em_fit(Y128, init, iterations=2, saliency=sal), Y128)
    assert np.abs(r['affiliation'] - ref).max() < 2e-4


def test_remainder_bin_members_equal_plain_launch_and_oracle():
    """513 bins on 256 CUs: bin 512 runs as eight member workgroups inside the grid (frame windows,
    partial sums through L2, distributed factorisation).  Same arithmetic per frame, a different
    summation order of the covariance sums: equal to the plain launch up to float32 rounding."""
    from oracle import cacgmm as oc, synth
    from pb_bss_amd import engine
    F, T, D, K = 513, 500, 8, 3
    Y, init = synth.make_stft(F, T, D, K, seed=0)
    r_split = _fit32(Y, init, 3)
    engine.set_split_tail(False)
    try:
        r_plain = _fit32(Y, init, 3)
    finally:
        engine.set_split_tail(True)
    assert engine.split_error() == 0
    assert np.abs(r_split['affiliation'][:512] - r_plain['affiliation'][:512]).max() == 0.0
    assert np.abs(r_split['affiliation'][512] - r_plain['affiliation'][512]).max() < 1e-4
    sel = [0, 100, 255, 256, 400, 511, 512]
    Y128 = Y[sel].astype(np.complex128)
    ref = oc.em_predict(oc.em_fit(Y128, init[sel], iterations=3), Y128)
    assert np.abs(r_split['affiliation'][sel] - ref).max() < 5e-4
    assert int(r_split['status'].max()) & 3 == 0


def test_statistical_tolerance_of_the_reference_test():
    """tests/test_distribution/test_cacgmm.py:24-56 of the reference, with fewer samples per
    mixture (the packed kernel keeps the frames in LDS) and several independent mixtures."""
    import pb_bss_amd
    from pb_bss_amd.distribution import CACGMMTrainer, sample_cacgmm
    np.random.seed(0)
    samples = 3000
    weight = np.array([0.3, 0.7])
    covariance = np.array(
        [[[10, 1 + 1j, 1 + 1j], [1 - 1j, 5, 1], [1 - 1j, 1, 2]],
         [[2, 0, 0], [0, 3, 0], [0, 0, 2]]], dtype=np.complex128)
    covariance /= np.trace(covariance, axis1=-2, axis2=-1)[..., None, None]
    x = np.stack([sample_cacgmm(samples, weight, covariance) for _ in range(4)]).astype(np.complex64)
    init = np.random.uniform(size=(4, 2, samples))
    init /= init.sum(axis=1, keepdims=True)
    with pb_bss_amd.arithmetic('reference'):
        model = CACGMMTrainer().fit(x, initialization=init, iterations=100, covariance_norm='trace')
    for b in range(4):
        cov = model.cacg.covariance[b]
        perm = min(itertools.permutations(range(2)),
                   key=lambda p: np.linalg.norm(cov[list(p)] - covariance))
        np.testing.assert_allclose(cov[list(perm)], covariance, atol=0.1)
        w = model.weight[b][list(perm)]
        assert w[0] < w[1], w
        np.testing.assert_allclose(w, weight[:, None], atol=0.15)


def test_trainer_switch_model_resume_and_fallbacks():
    import pb_bss_amd
    from pb_bss_amd.distribution import CACGMMTrainer
    from oracle import cacgmm as oc, synth
    Y, init = synth.make_stft(6, 200, 4, 3, seed=8)
    Y128 = Y.astype(np.complex128)
    ref = oc.em_predict(oc.em_fit(Y128, init, iterations=3), Y128)
    with pb_bss_amd.arithmetic('reference'):
        m32 = CACGMMTrainer().fit(Y, initialization=init, iterations=3)
        got = m32.predict(Y)
        # complex128 input: the reference computes in float64 -> float64 kernel (exact parity)
        m64 = CACGMMTrainer().fit(Y128, initialization=init, iterations=3)
        assert np.abs(m64.predict(Y128) - ref).max() < 1e-9
        # K = 7 is not served by the packed kernel -> generic-size float64 path, silently
        Y7, init7 = synth.make_stft(3, 120, 4, 7, seed=9)
        CACGMMTrainer().fit(Y7, initialization=init7, iterations=2)
    assert np.abs(got - ref).max() < 5e-4
    assert 1e-9 < np.abs(got - ref).max()  # it really was the single-precision kernel
    # resume from a model through the C ABI (precision F32, model initialisation)
    from pb_bss_amd import _lib, engine
    r1 = engine.em_fit(_lib.to_device(Y), 3, gamma0=_lib.to_device(init), iterations=2,
                       precision='f32')
    r2 = engine.em_fit(_lib.to_device(Y), 3, model=(r1['eigvec'], r1['eigval'], r1['weight']),
                       iterations=1, final_predict=True, precision='f32')
    assert np.abs(_lib.to_host(r2['affiliation']) - ref).max() < 5e-4
    # refusals of the C entry point
    with pytest.raises(NotImplementedError):
        engine.em_fit(_lib.to_device(Y128), 3, gamma0=_lib.to_device(init), iterations=1,
                      precision='f32')
    with pytest.raises(NotImplementedError):
        engine.em_fit(_lib.to_device(Y7), 7, gamma0=_lib.to_device(init7), iterations=1,
                      precision='f32')
