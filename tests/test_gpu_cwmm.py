"""GPU: complex-Watson mixture model (SURVEY 8f row N2 / BASELINE config 4)
against vectors of the real reference and the NumPy oracle."""
import ast
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def cos_sim(a, b):
    return np.abs(np.einsum('...d,...d->...', a.conj(), b)) / (
        np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize('name', ['cwmm_f6_t120_d6_k3', 'cwmm_f5_t90_d3_k2',
                                  'cwmm_f4_t80_d8_k2_uniform'])
def test_cwmm_trainer_matches_reference(name):
    from pb_bss_amd.distribution import CWMMTrainer
    g = load(name)
    kw = ast.literal_eval(str(g['kwargs']))
    model = CWMMTrainer().fit(g['Y'], initialization=g['init'],
                              iterations=int(g['iterations']), **kw)
    assert model.weight.shape == g['weight'].shape
    assert model.complex_watson.mode.shape == g['mode'].shape
    assert model.complex_watson.concentration.shape == g['concentration'].shape
    assert np.abs(model.weight - g['weight']).max() < 1e-9
    assert np.abs(model.complex_watson.concentration - g['concentration']).max() \
        < 1e-8 * g['concentration'].max()
    assert np.abs(cos_sim(model.complex_watson.mode, g['mode']) - 1).max() < 1e-10
    aff = model.predict(g['Y'])
    assert np.abs(aff - g['affiliation']).max() < 1e-8
    yn = g['Y'].astype(np.complex128)
    yn = yn / np.linalg.norm(yn, axis=-1, keepdims=True)
    lp = model.complex_watson.log_pdf(yn[..., None, :, :])
    assert lp.shape == g['log_pdf'].shape
    assert np.abs(lp - g['log_pdf']).max() < 1e-7 * np.abs(g['log_pdf']).max()


def test_cwmm_config4_shape_and_mvdr():
    """BASELINE config 4 (reduced F): 6-mic array, T=800, K=3 through the Watson
    trainer, then MVDR-Souden extraction, against the oracle chain."""
    from pb_bss_amd.distribution import CWMMTrainer
    from pb_bss_amd import extraction as ex
    from oracle import beamformer as ob, cwmm as ow, synth
    F, T, D, K = 33, 800, 6, 3
    Y, init = synth.make_stft(F, T, D, K, seed=4)
    Y128 = Y.astype(np.complex128)
    masks = CWMMTrainer().fit_predict(Y, initialization=init, iterations=15)
    ref = ow.cwmm_predict(ow.cwmm_fit(Y128, init, iterations=15), Y128)
    assert masks.shape == (F, K, T)
    assert np.abs(masks - ref).max() < 1e-7
    X = Y128.transpose(0, 2, 1)
    psd = ex.get_power_spectral_density_matrix(Y.transpose(0, 2, 1), masks)
    psd_ref = ob.psd(X, ref)
    w = ex.get_mvdr_vector_souden(psd[:, 0], psd[:, 1] + psd[:, 2])
    w_ref = ob.mvdr_souden(psd_ref[:, 0], psd_ref[:, 1] + psd_ref[:, 2])
    assert np.abs(w - w_ref).max() < 1e-6 * np.abs(w_ref).max()


def test_watson_trainer_and_scalar_functions():
    from pb_bss_amd.distribution import ComplexWatsonTrainer, CWMMTrainer
    from oracle import cwmm as ow
    g = load('watson_scalar_functions')
    t = ComplexWatsonTrainer(5)
    assert np.allclose(t.hypergeometric_ratio_inverse(g['eigenvalues']), g['ratio_inverse_d5'],
                       rtol=1e-13, atol=0)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((3, 400, 5)) + 1j * rng.standard_normal((3, 400, 5))
    y[..., 0] *= 3
    m = t.fit(y)
    yn = ow.normalize_observation(y)
    mode, conc = ow.watson_m_step(yn, np.ones((3, 400)), ow.make_spline(5))
    assert np.abs(m.concentration - conc).max() < 1e-9 * conc.max()
    assert np.abs(cos_sim(m.mode, mode) - 1).max() < 1e-12
    # reference test (tests/test_distribution/test_cwmm.py:33-37): shapes
    np.random.seed(0)
    x = rng.standard_normal((2000, 3)) + 1j * rng.standard_normal((2000, 3))
    model = CWMMTrainer().fit(x, num_classes=2, iterations=10)
    assert model.weight.shape == (2, 1)
    assert model.complex_watson.mode.shape == (2, 3)
    assert model.complex_watson.concentration.shape == (2,)
    with pytest.raises(AssertionError):
        CWMMTrainer().fit(x, num_classes=2, affiliation_eps=1e-10)


def test_cwmm_stepwise_shared_weights():
    from pb_bss_amd.distribution import CWMMTrainer
    from oracle import cwmm as ow, synth
    Y, init = synth.make_stft(5, 70, 4, 2, seed=12)
    Y128 = Y.astype(np.complex128)
    m = CWMMTrainer().fit(Y, initialization=init, iterations=3, weight_constant_axis=(-3, -1))
    ref = ow.cwmm_fit(Y128, init, iterations=3, weight_constant_axis=(-3, -1))
    assert m.weight.shape == ref['weight'].shape == (1, 2, 1)
    assert np.abs(m.weight - ref['weight']).max() < 1e-10
    assert np.abs(m.complex_watson.concentration - ref['concentration']).max() < 1e-8


@pytest.mark.parametrize('with_sal', [False, True])
def test_cwmm_frame_varying_weights_on_the_device(with_sal):
    """weight_constant_axis=(-3,): weights (1, K, N) shared by the bins but varying over the
    frames (reference cwmm.py:151-182 with mixture_model_utils.py:184-201) -- the step-wise loop
    stays on the device (log-pdf kernel + pbbss_log_pdf_to_affiliation + weight reduction)."""
    from pb_bss_amd.distribution import CWMMTrainer
    from oracle import cwmm as ow, synth
    Y, init = synth.make_stft(7, 90, 5, 3, seed=14)
    Y128 = Y.astype(np.complex128)
    sal = np.random.default_rng(2).uniform(0.3, 1.0, size=(7, 90)) if with_sal else None
    m = CWMMTrainer().fit(Y, initialization=init, iterations=4, weight_constant_axis=(-3,),
                          saliency=sal)
    ref = ow.cwmm_fit(Y128, init, iterations=4, weight_constant_axis=(-3,), saliency=sal)
    assert m.weight.shape == ref['weight'].shape == (1, 3, 90)
    assert np.abs(m.weight - ref['weight']).max() < 1e-10
    assert np.abs(m.complex_watson.concentration - ref['concentration']).max() < 1e-7
    assert np.abs(m.predict(Y) - ow.cwmm_predict(ref, Y128)).max() < 1e-8


@pytest.mark.parametrize('F,T,D,K,with_sal', [(257, 200, 4, 2, False), (259, 150, 6, 3, True)])
def test_cwmm_remainder_bins_as_split_groups(F, T, D, K, with_sal):
    """2^n + 1 bins: the bins beyond a multiple of the CU count run as split groups on the side
    stream (csrc/cwmm.hpp: WatsonSplit) -- same model as the one-launch fit
    (pbbss_set_split_tail(0)) to rounding of the frame sums, and as the oracle."""
    from pb_bss_amd import engine
    from pb_bss_amd.distribution import CWMMTrainer
    from oracle import cwmm as ow, synth
    Y, init = synth.make_stft(F, T, D, K, seed=F + D)
    Y128 = Y.astype(np.complex128)
    sal = np.random.default_rng(F).uniform(0.2, 1.0, size=(F, T)) if with_sal else None
    ref = ow.cwmm_fit(Y128, init, iterations=6, saliency=sal)
    want = ow.cwmm_predict(ref, Y128)
    got = {}
    try:
        for split in (1, 0):
            engine.set_split_tail(split)
            got[split] = CWMMTrainer().fit_predict(Y, initialization=init, iterations=6, saliency=sal)
    finally:
        engine.set_split_tail(1)
    assert np.abs(got[1] - want).max() < 1e-7
    assert np.abs(got[1] - got[0]).max() < 1e-9
    assert np.abs(got[1][256:] - want[256:]).max() < 1e-7  # the split bins themselves


def test_cwmm_config4_full_size():
    """BASELINE configs[3] at its full size: 6-mic array, F = 257, T = 800, K = 3 (257 = 256 + 1
    bins: the last one runs as split groups), Watson trainer -> MVDR-Souden, against the oracle."""
    from pb_bss_amd.distribution import CWMMTrainer
    from pb_bss_amd import extraction as ex
    from oracle import beamformer as ob, cwmm as ow, synth
    F, T, D, K = 257, 800, 6, 3
    Y, init = synth.make_stft(F, T, D, K, seed=14)
    Y128 = Y.astype(np.complex128)
    masks = CWMMTrainer().fit_predict(Y, initialization=init, iterations=12)
    ref = ow.cwmm_predict(ow.cwmm_fit(Y128, init, iterations=12), Y128)
    assert masks.shape == (F, K, T)
    assert np.abs(masks - ref).max() < 1e-6
    assert np.abs(masks[256] - ref[256]).max() < 1e-6
    psd = ex.get_power_spectral_density_matrix(Y.transpose(0, 2, 1), masks)
    psd_ref = ob.psd(Y128.transpose(0, 2, 1), ref)
    w = ex.get_mvdr_vector_souden(psd[:, 0], psd[:, 1] + psd[:, 2])
    w_ref = ob.mvdr_souden(psd_ref[:, 0], psd_ref[:, 1] + psd_ref[:, 2])
    assert np.abs(w - w_ref).max() < 1e-5 * np.abs(w_ref).max()


@pytest.mark.parametrize('T', [40, 64, 65, 129, 256, 257, 330])
def test_cwmm_frame_counts_around_the_chunk_size(T):
    """The Watson kernel shares the 64-frame chunk layout of the LDS frame arrays (round 5: unmasked
    M sweep, E passes that skip empty chunks, padding weights rewritten with zeros): frame counts on
    both sides of the chunk and pass boundaries, with and without saliency (zeros included), and the
    PSD kernel on the same layout (masked, mask-less, un-normalised)."""
    from pb_bss_amd.distribution import CWMMTrainer
    from pb_bss_amd import extraction as ex
    from oracle import beamformer as ob, cwmm as ow, synth
    rng = np.random.default_rng(T)
    for D, K in ((3, 2), (6, 3), (8, 4)):
        F = 3
        Y, init = synth.make_stft(F, T, D, K, seed=T + D)
        Y128 = Y.astype(np.complex128)
        sal = rng.uniform(0.0, 1.0, size=(F, T))
        sal[:, ::5] = 0.0
        for s in (None, sal):
            ref = ow.cwmm_fit(Y128, init, iterations=4, saliency=s)
            got = CWMMTrainer().fit_predict(Y, initialization=init, iterations=4, saliency=s)
            assert np.abs(got - ow.cwmm_predict(ref, Y128)).max() < 1e-7, (T, D, K, s is not None)
        X, X128 = Y.transpose(0, 2, 1), Y128.transpose(0, 2, 1)
        np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X, got),
                                   ob.psd(X128, got), atol=1e-11)
        np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X, got, normalize=False),
                                   ob.psd(X128, got, normalize=False), atol=1e-9)
        np.testing.assert_allclose(ex.get_power_spectral_density_matrix(X),
                                   ob.psd(X128), atol=1e-11)
