"""GPU: the end-to-end batch pipeline of BASELINE config 3 (reduced size):
cACGMM EM on a batch -> DHTV alignment -> PSD -> 'gev+ban' -> apply, all on the
device, against the same chain built from the NumPy oracles."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))


def test_batch_pipeline_matches_oracle_chain():
    from separate_batch import separate
    from oracle import beamformer as ob, cacgmm as oc, permutation_alignment as op, synth
    from pb_bss_amd import _lib
    U, F, T, D, K, iters = 2, 257, 90, 4, 2, 8
    data = [synth.make_stft(F, T, D, K, seed=40 + u) for u in range(U)]
    Y = np.stack([d[0] for d in data])
    init = np.stack([d[1] for d in data])
    out = separate(Y, init, iters, stft_size=512)
    masks = _lib.to_host(out['masks'])
    enhanced = _lib.to_host(out['enhanced'])
    assert masks.shape == (U, K, F, T) and enhanced.shape == (U, K, F, T)
    plan = op.alignment_plan(512, **op.PRESETS[512])
    for u in range(U):
        Y128 = Y[u].astype(np.complex128)
        ref = oc.em_predict(oc.em_fit(Y128, init[u], iterations=iters), Y128)  # (F, K, T)
        kft = ref.transpose(1, 0, 2)
        mapping = op.dhtv_calculate_mapping(kft, plan)
        assert (mapping == _lib.to_host(out['mapping'])[u]).all()
        aligned = op.apply_mapping(kft, mapping)
        assert np.abs(aligned - masks[u]).max() < 1e-9
        X = Y128.transpose(0, 2, 1)
        psd = ob.psd(X, aligned.transpose(1, 0, 2))
        for k in range(K):
            w = ob.bf_vector('gev+ban', psd[:, k], psd.sum(1) - psd[:, k])
            s = ob.apply_bf(w, X)
            # GEV vectors carry an arbitrary phase per frequency: compare magnitudes
            assert np.abs(np.abs(s) - np.abs(enhanced[u, k])).max() < 1e-7 * np.abs(s).max()


def _phase_aligned_error(got, ref):
    """max |got e^{-i phi} - ref| / max |ref| with one phase per (class, bin): a GEV
    beamformer is defined up to a unit-modulus factor per frequency (LAPACK's choice of
    eigenvector phase), so the complex outputs are compared after removing exactly that."""
    inner = (got * np.conj(ref)).sum(-1, keepdims=True)
    phase = inner / np.maximum(np.abs(inner), 1e-300)
    return np.abs(got * np.conj(phase) - ref).max() / np.abs(ref).max()


def test_config3_chain_full_size_utterances_match_oracle_chain():
    """BASELINE configs[2] at its real per-utterance size (F=513, T=500, D=8, K=3): four
    utterances through pb_bss_amd.pipeline.separate (EM -> DHTV -> PSD -> gev+ban -> apply)
    against the same chain assembled from the NumPy oracles -- masks, permutation mapping,
    beamformer (|cos|) and the COMPLEX enhanced signals up to the per-bin GEV phase."""
    from oracle import beamformer as ob, cacgmm as oc, permutation_alignment as op, synth
    from pb_bss_amd import _lib, pipeline
    U, F, T, D, K, iters = 4, 513, 500, 8, 3, 20
    data = [synth.make_stft(F, T, D, K, seed=u) for u in range(U)]
    Y = np.stack([d[0] for d in data])
    init = np.stack([d[1] for d in data])
    out = pipeline.separate(_lib.to_device(Y), _lib.to_device(init), iters, 1024)
    masks = _lib.to_host(out['masks'])
    enhanced = _lib.to_host(out['enhanced'])
    w_dev = _lib.to_host(out['bf_vector'])
    mapping_dev = _lib.to_host(out['mapping'])
    assert masks.shape == (U, K, F, T) and enhanced.shape == (U, K, F, T)
    plan = op.alignment_plan(1024, **op.PRESETS[1024])
    for u in range(U):
        Y128 = Y[u].astype(np.complex128)
        ref = oc.em_predict(oc.em_fit(Y128, init[u], iterations=iters), Y128)  # (F, K, T)
        kft = ref.transpose(1, 0, 2)
        mapping = op.dhtv_calculate_mapping(kft, plan)
        assert (mapping == mapping_dev[u]).all()
        aligned = op.apply_mapping(kft, mapping)
        assert np.abs(aligned - masks[u]).max() < 1e-9
        X = Y128.transpose(0, 2, 1)
        psd = ob.psd(X, aligned.transpose(1, 0, 2))
        for k in range(K):
            w = ob.bf_vector('gev+ban', psd[:, k], psd.sum(1) - psd[:, k])
            cos = np.abs((np.conj(w) * w_dev[u, k]).sum(-1)) / (
                np.linalg.norm(w, axis=-1) * np.linalg.norm(w_dev[u, k], axis=-1))
            assert np.abs(cos - 1).max() < 1e-9
            s = ob.apply_bf(w, X)
            assert _phase_aligned_error(enhanced[u, k], s) < 1e-8


def test_pipeline_with_mvdr_souden_matches_oracle_chain():
    """pipeline.separate(beamformer='mvdr_souden'): EM -> DHTV -> PSD -> MVDR-Souden with the
    automatic reference channel (beamformer.py:627-698, :601-624; the batched device stage picks
    the channel per (class, utterance) from the SNR summed over all bins) -> apply.  MVDR has no
    phase ambiguity: the complex outputs are compared directly."""
    from oracle import beamformer as ob, cacgmm as oc, permutation_alignment as op, synth
    from pb_bss_amd import _lib, pipeline
    U, F, T, D, K, iters = 3, 257, 120, 5, 3, 6
    data = [synth.make_stft(F, T, D, K, seed=60 + u) for u in range(U)]
    Y = np.stack([d[0] for d in data])
    init = np.stack([d[1] for d in data])
    out = pipeline.separate(_lib.to_device(Y), _lib.to_device(init), iters, 512,
                            beamformer='mvdr_souden')
    w_dev = _lib.to_host(out['bf_vector'])       # (U, K, F, D)
    s_dev = _lib.to_host(out['enhanced'])        # (U, K, F, T)
    plan = op.alignment_plan(512, **op.PRESETS[512])
    for u in range(U):
        Y128 = Y[u].astype(np.complex128)
        ref = oc.em_predict(oc.em_fit(Y128, init[u], iterations=iters), Y128)
        kft = ref.transpose(1, 0, 2)
        mapping = op.dhtv_calculate_mapping(kft, plan)
        assert (mapping == _lib.to_host(out['mapping'])[u]).all()
        aligned = op.apply_mapping(kft, mapping)
        X = Y128.transpose(0, 2, 1)
        psd = ob.psd(X, aligned.transpose(1, 0, 2))
        for k in range(K):
            w = ob.mvdr_souden(psd[:, k], psd.sum(1) - psd[:, k])
            assert np.abs(w - w_dev[u, k]).max() < 1e-7 * np.abs(w).max()
            s = ob.apply_bf(w, X)
            assert np.abs(s - s_dev[u, k]).max() < 1e-7 * np.abs(s).max()
    with pytest.raises(ValueError):
        pipeline.separate(_lib.to_device(Y), _lib.to_device(init), 2, 512, beamformer='lcmv')


def test_single_utterance_without_the_leading_axis():
    from oracle import synth
    from pb_bss_amd import _lib, pipeline
    Y, init = synth.make_stft(257, 60, 4, 2, seed=7)
    yd, gd = _lib.to_device(Y), _lib.to_device(init)
    for bf in pipeline.BEAMFORMERS:
        one = pipeline.separate(yd, gd, 5, 512, beamformer=bf)
        ref = pipeline.separate(yd[None], gd[None], 5, 512, beamformer=bf)
        assert tuple(one['masks'].shape) == (2, 257, 60) and tuple(one['mapping'].shape) == (2, 257)
        for k in ref:
            assert (one[k] == ref[k][0]).all(), k


def test_reference_channel_selection_stays_on_the_device_and_defers_its_assert():
    """pipeline.device_ops.mvdr_souden (round 6): the SNR-optimal reference channel
    (beamformer.py:601-624) is chosen by device ops -- same channel as the host selection of
    get_mvdr_vector_souden, first maximum on ties --, the reference's finiteness assert is
    raised by device_ops.assert_finite() instead of in the middle of the step."""
    import torch
    from pb_bss_amd import _lib
    from pb_bss_amd import extraction as ex
    from pb_bss_amd.extraction.beamformer import (_select_reference_channel,
                                                  _select_reference_channel_device)
    from pb_bss_amd.pipeline import device_ops as ops
    rng = np.random.default_rng(5)
    K, F, D = 3, 40, 5

    def psd(*lead):
        a = rng.standard_normal((*lead, D, 2 * D)) + 1j * rng.standard_normal((*lead, D, 2 * D))
        return a @ a.conj().swapaxes(-1, -2) / (2 * D)
    target, noise = psd(K, F), psd(K, F) + 0.1 * np.eye(D)
    ops.assert_finite()                                   # start from an empty queue
    w = _lib.to_host(ops.mvdr_souden(_lib.to_device(target), _lib.to_device(noise)))
    ops.assert_finite()
    for k in range(K):
        want, ref = ex.get_mvdr_vector_souden(target[k], noise[k], return_ref_channel=True)
        np.testing.assert_allclose(w[k], want, rtol=1e-12, atol=1e-14)
    # ties: the first maximum, like np.argmax
    num = torch.ones((2, 7, 4), dtype=torch.complex128, device='cuda')
    den = torch.ones((2, 7, 4), dtype=torch.complex128, device='cuda')
    num[1, :, 2] = 3.0
    num[1, :, 3] = 3.0
    idx, ok = _select_reference_channel_device(num, den, 1e-300)
    assert idx.tolist() == [0, 2] and bool(ok)
    assert [_select_reference_channel(_lib.to_host(num[i]), _lib.to_host(den[i]), 1e-300)
            for i in range(2)] == [0, 2]
    bad = target.copy()
    bad[1, 3] = np.nan
    ops.mvdr_souden(_lib.to_device(bad), _lib.to_device(noise))   # returns: nothing is read back
    with pytest.raises(AssertionError):
        ops.assert_finite()
    ops.assert_finite()                                   # the queue is empty again



@pytest.mark.gpu
def test_select_reference_channel_kernel_against_the_host_selection():
    """pbbss_select_reference_channel (round 6): sums over the bins, NumPy's complex maximum with
    eps, first arg-max of the real part, column gather and the finiteness flag in one launch --
    against `_select_reference_channel` (the host statement of beamformer.py:616-624) on both
    problem layouts ((L, F) and (F, L) order of the matrices), with ties, denominators below eps,
    a complex denominator whose real part equals eps, D up to 32 and a NaN."""
    import torch
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.extraction.beamformer import _select_reference_channel
    rng = np.random.default_rng(11)
    for L, F, D in ((3, 257, 6), (1, 5, 2), (4, 70, 32), (2, 1, 8)):
        mat = rng.standard_normal((L, F, D, D)) + 1j * rng.standard_normal((L, F, D, D))
        num = rng.uniform(0.5, 2.0, (L, F, D)) + 1e-3j * rng.standard_normal((L, F, D))
        den = rng.uniform(0.5, 2.0, (L, F, D)) + 1e-3j * rng.standard_normal((L, F, D))
        eps = 1e-300
        if L > 1:
            den[1, :, 0] = 0.0                       # below eps: the floor wins
            eps = 1e-3
            num[L - 1, :, D - 1] = num[L - 1, :, D - 2]   # a tie: the first maximum
            den[L - 1, :, D - 1] = den[L - 1, :, D - 2]
        for order in ('lf', 'fl'):
            if order == 'lf':
                m, n, d = (_lib.to_device(a.reshape(L * F, *a.shape[2:])) for a in (mat, num, den))
                w, ref, ok = engine.select_reference_channel(m, n, d, L, F, eps)
            else:
                m, n, d = (_lib.to_device(np.ascontiguousarray(a.swapaxes(0, 1)).reshape(
                    L * F, *a.shape[2:])) for a in (mat, num, den))
                w, ref, ok = engine.select_reference_channel(m, n, d, L, F, eps, lead_stride=1,
                                                             bin_stride=L)
            want = [_select_reference_channel(num[i], den[i], eps) for i in range(L)]
            assert ref.tolist() == want, (L, F, D, order)
            assert ok.tolist() == [1] * L
            got = _lib.to_host(w)
            for i in range(L):
                np.testing.assert_array_equal(got[i], mat[i, :, :, want[i]])
    # the complex maximum keeps a denominator whose real part EQUALS eps and whose imaginary part
    # is non-negative (np.maximum orders complex numbers by real part, then imaginary part)
    num = torch.ones((1, 1, 2), dtype=torch.complex128, device='cuda')
    den = torch.tensor([[[0.5 + 0.5j, 0.5 - 0.5j]]], dtype=torch.complex128, device='cuda')
    mat = torch.zeros((1, 2, 2), dtype=torch.complex128, device='cuda')
    _, ref, ok = engine.select_reference_channel(mat, num.reshape(1, 2), den.reshape(1, 2), 1, 1, 0.5)
    hn, hd = _lib.to_host(num[0]), _lib.to_host(den[0])
    assert ref.tolist() == [int(np.argmax((hn.sum(0) / np.maximum(hd.sum(0), 0.5)).real))]
    # an all-zero class (silent source) with the floor eps = tiny: 0 / tiny = 0, finite -- the
    # quotient must not go through |den|^2 (tiny^2 underflows to 0)
    tiny = float(np.finfo(np.float64).tiny)
    zn = torch.zeros((2, 3), dtype=torch.complex128, device='cuda')
    zn[:, 1] = 1.0
    zd = torch.zeros((2, 3), dtype=torch.complex128, device='cuda')
    _, ref, ok = engine.select_reference_channel(
        torch.zeros((2, 3, 3), dtype=torch.complex128, device='cuda'), zn, zd, 1, 2, tiny)
    hs = _lib.to_host(zn).sum(0) / np.maximum(_lib.to_host(zd).sum(0), tiny)
    assert ok.tolist() == [int(np.all(np.isfinite(hs)))] and ref.tolist() == [int(np.argmax(hs.real))]
    # a NaN: reported, and it is the arg-max (np.argmax returns the first NaN)
    bad = torch.ones((1, 3, 4), dtype=torch.complex128, device='cuda')
    bad[0, 1, 2] = float('nan')
    _, ref, ok = engine.select_reference_channel(
        torch.zeros((3, 4, 4), dtype=torch.complex128, device='cuda'), bad.reshape(3, 4),
        torch.ones((3, 4), dtype=torch.complex128, device='cuda'), 1, 3, 1e-300)
    assert ok.tolist() == [0] and ref.tolist() == [2]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [np.complex64, np.complex128])
def test_apply_beamforming_vector_shares_the_observation_over_leading_axes(dtype):
    """apply_beamforming_vector(vector (K, F, D), mix (F, D, T)) -- the broadcast of the reference's
    einsum '...a,...at->...t' (beamformer.py:572-583) -- reads the one observation K times
    (pbbss_apply_beamforming_vector_shared) instead of expanding it; also (2, K, F, D) on (F, D, T)
    and the plain same-shape call."""
    from pb_bss_amd import extraction as ex
    rng = np.random.default_rng(3)
    K, F, D, T = 3, 17, 5, 130
    x = (rng.standard_normal((F, D, T)) + 1j * rng.standard_normal((F, D, T))).astype(dtype)
    for lead in ((K,), (2, K), ()):
        w = rng.standard_normal((*lead, F, D)) + 1j * rng.standard_normal((*lead, F, D))
        got = ex.apply_beamforming_vector(w, x)
        want = np.einsum('...a,...at->...t', w.conj(), x.astype(np.complex128))
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_graphed_extraction_stage_replays_bit_identically():
    """pipeline.graphed: PSD -> MVDR-Souden (device-side reference channel) -> apply captured once
    into a HIP graph; replays on NEW masks give exactly what the eager stage gives on them, the
    deferred finiteness check still fires for a replay with a NaN, and a stage that synchronises
    with the host (gev: status read-back) falls back to eager calls with a RuntimeWarning."""
    import torch
    from pb_bss_amd import _lib
    from pb_bss_amd.pipeline import device_ops as ops, graphed
    rng = np.random.default_rng(8)
    F, T, D, K = 33, 120, 4, 3
    X = _lib.to_device((rng.standard_normal((F, D, T)) + 1j * rng.standard_normal((F, D, T)))
                       .astype(np.complex64))

    def masks():
        m = rng.uniform(0.05, 1.0, size=(F, K, T))
        return _lib.to_device(m / m.sum(1, keepdims=True))

    def extract(mk):
        psd = ops.psd(X, mk)
        target = psd.movedim(-3, 0).contiguous()
        noise = (psd.sum(dim=-3).unsqueeze(0) - target).contiguous()
        w = ops.mvdr_souden(target, noise)
        return w, ops.apply_bf(w, X)

    ops.assert_finite()
    g = graphed(extract, masks())
    assert g.captured
    for _ in range(3):
        mk = masks()
        w, s = g(mk)
        w, s = w.clone(), s.clone()                    # the results are the graph's own buffers
        we, se = extract(mk)
        assert torch.equal(w, we) and torch.equal(s, se)
    ops.assert_finite()
    bad = masks()
    bad[3, 1, :] = float('nan')
    g(bad)
    with pytest.raises(AssertionError):
        ops.assert_finite()
    ops.assert_finite()

    def with_host_sync(mk):
        psd = ops.psd(X, mk)
        return ops.gev_ban(psd[:, 0].contiguous(), (psd.sum(1) - psd[:, 0]).contiguous())
    with pytest.warns(RuntimeWarning, match='capture failed'):
        h = graphed(with_host_sync, masks())
    assert not h.captured
    mk = masks()
    assert torch.equal(h(mk), with_host_sync(mk))
