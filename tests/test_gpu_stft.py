"""STFT edge: HIP stft / istft against the NumPy oracle
(oracle/stft.py, the restated nara_wpe.utils algorithm) and through size-independent
properties.  Tolerances: float64 FFT, 1e-11 absolute on O(1) signals (unit-variance noise
times a window <= 1: bins are O(sqrt(size)))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _signal(shape, seed=0, dtype=np.float64):
    return np.random.default_rng(seed).standard_normal(shape).astype(dtype)


@pytest.mark.parametrize('size,shift,n', [(512, 128, 8000), (1024, 256, 16000), (64, 16, 1000),
                                          (256, 64, 255), (2048, 512, 5000), (8, 2, 37),
                                          (4096, 1024, 9000)])
def test_stft_matches_oracle(size, shift, n):
    from oracle import stft as o
    from pb_bss_amd.transform import stft
    x = _signal((3, n), seed=size)
    ref = o.stft(x, size, shift)
    got = stft(x, size, shift)
    assert got.shape == ref.shape and got.dtype == np.complex128
    np.testing.assert_allclose(got, ref, atol=1e-11)


@pytest.mark.parametrize('kw', [dict(fading=False), dict(pad=False), dict(fading=False, pad=False),
                                dict(window_length=400, ), dict(window='hann'),
                                dict(window='hamming', symmetric_window=True),
                                dict(window=np.bartlett)])
def test_stft_options(kw):
    from oracle import stft as o
    from pb_bss_amd.transform import stft
    x = _signal((2, 4321), seed=3)
    size, shift = 512, (100 if 'window_length' in kw else 128)
    ref = o.stft(x, size, shift, **kw)
    got = stft(x, size, shift, **kw)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=1e-11)


def test_stft_inputs_axes_layout_and_dtype():
    import torch
    from einops import rearrange
    from oracle import stft as o
    from pb_bss_amd.transform import stft
    x = _signal((2, 3, 3000), seed=5)
    ref = o.stft(x, 512, 128)
    np.testing.assert_allclose(stft(x, 512, 128), ref, atol=1e-11)        # leading axes
    np.testing.assert_allclose(stft(x[0, 0], 512, 128), ref[0, 0], atol=1e-11)  # 1-D
    xt = np.moveaxis(x, -1, 1)                                            # (2, 3000, 3)
    got = stft(xt, 512, 128, axis=1)                                      # frames at 1, bins at 2
    np.testing.assert_allclose(got, np.moveaxis(ref, (-2, -1), (1, 2)), atol=1e-11)
    x32 = x[0].astype(np.float32)                                         # float32 audio
    ref32 = o.stft(x32.astype(np.float64), 512, 128)
    np.testing.assert_allclose(stft(x32, 512, 128), ref32, atol=1e-11)
    got64 = stft(x32, 512, 128, dtype=np.complex64)
    assert got64.dtype == np.complex64
    np.testing.assert_allclose(got64, ref32, atol=2e-5)
    # the arrangement of test_spatial_mm.py:41, written by the kernel
    ftd = stft(x[0], 512, 128, layout='f t d')
    np.testing.assert_array_equal(ftd, rearrange(stft(x[0], 512, 128), 'd t f -> f t d'))
    # torch CUDA in -> torch CUDA out on the same device
    xt = torch.from_numpy(x[0]).cuda()
    out = stft(xt, 512, 128)
    assert out.is_cuda and out.dtype == torch.complex128
    np.testing.assert_allclose(out.cpu().numpy(), ref[0], atol=1e-11)
    # int16 PCM is converted like numpy would
    pcm = (x[0, 0] * 1000).astype(np.int16)
    np.testing.assert_allclose(stft(pcm, 512, 128), o.stft(pcm.astype(np.float64), 512, 128),
                               atol=1e-8)


@pytest.mark.parametrize('size,shift,kw', [(512, 128, {}), (1024, 256, {}), (512, 128, dict(fading=False)),
                                           (256, 64, dict(window='hann')),
                                           (512, 100, dict(window_length=400))])
def test_istft_matches_oracle(size, shift, kw):
    from oracle import stft as o
    from pb_bss_amd.transform import istft, stft_frames_to_samples
    rng = np.random.default_rng(size + shift)
    T, F = 37, size // 2 + 1
    X = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))  # not a valid STFT
    ref = o.istft(X, size, shift, **kw)
    got = istft(X, size, shift, **kw)
    assert got.shape == ref.shape == (2, stft_frames_to_samples(T, size, shift, **{
        k: v for k, v in kw.items() if k in ('fading', 'window_length')}))
    np.testing.assert_allclose(got, ref, atol=1e-12)
    got32 = istft(X.astype(np.complex64), size, shift, **kw)
    # (numpy >= 2 would run a complex64 irfft in single precision; the device always uses float64)
    np.testing.assert_allclose(got32, o.istft(X.astype(np.complex64).astype(np.complex128), size, shift, **kw),
                               atol=1e-12)
    np.testing.assert_allclose(istft(X[0, :], size, shift, **kw), ref[0], atol=1e-12)  # (T, F)


def test_round_trip_at_full_size_and_reference_call_site():
    """istft(stft(x))[..., :N] == x: the call pattern of test_spatial_mm.py:17-22 on a
    CHiME-sized utterance (6 channels, 8 s at 16 kHz)."""
    from pb_bss_amd.transform import stft, istft
    x = _signal((6, 128000), seed=9)
    X = stft(x, 1024, 256)
    assert X.shape == (6, 503, 513)
    y = istft(X, 1024, 256)[..., :x.shape[-1]]
    np.testing.assert_allclose(y, x, atol=1e-12)
    X = stft(x, 512, 128)
    np.testing.assert_allclose(istft(X, 512, 128)[..., :x.shape[-1]], x, atol=1e-12)
    # Parseval with the periodic Hann window at 75 % overlap (sum_m w^2[n + m shift] = 1.5):
    # size-independent check of the forward transform alone
    Xh = stft(x[:1], 1024, 256, window='hann')
    e_f = (np.abs(Xh[..., 1:-1]) ** 2).sum() * 2 + (np.abs(Xh[..., [0, -1]]) ** 2).sum()
    np.testing.assert_allclose(e_f / 1024, 1.5 * (x[:1] ** 2).sum(), rtol=1e-10)


def test_known_answer_and_linearity():
    from pb_bss_amd.transform import stft
    n = np.arange(4096)
    x = np.cos(2 * np.pi * 32 * n / 512)          # exactly bin 32 of a 512-point frame
    X = stft(x, 512, 128, window=np.ones, fading=False)
    mag = np.abs(X)
    assert np.all(np.argmax(mag, axis=-1) == 32)
    np.testing.assert_allclose(mag[:, 32], 256.0, atol=1e-9)
    mag[:, 32] = 0
    assert mag.max() < 1e-9
    a, b = _signal((2, 3000), seed=1)
    np.testing.assert_allclose(stft(2.0 * a - 3.0 * b, 512, 128),
                               2.0 * stft(a, 512, 128) - 3.0 * stft(b, 512, 128), atol=1e-11)


def test_errors():
    from pb_bss_amd.transform import stft, istft
    with pytest.raises(NotImplementedError):
        stft(np.zeros(1000), 500, 125)            # size not a power of two
    with pytest.raises(ValueError):
        istft(np.zeros((4, 257), complex), 512, 100)   # window_length % shift != 0
    with pytest.raises(AssertionError):
        istft(np.zeros((4, 256), complex), 512, 128)
    with pytest.raises(ValueError):
        stft(np.zeros((2, 3, 1000)), 512, 128, layout='f t d')


def test_audio_to_audio_separation_stays_on_device():
    """stft (f t d) -> cACGMM -> mask the reference channel -> istft, torch tensors throughout."""
    import torch
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.transform import stft, istft
    rng = np.random.default_rng(4)
    N, D, K = 16000, 4, 2
    src = rng.standard_normal((K, N)) * (np.sin(np.arange(N)[None] * np.array([[3e-3], [7e-3]])) > 0)
    mix = np.einsum('dk,kn->dn', rng.standard_normal((D, K)), src) + 0.01 * rng.standard_normal((D, N))
    x = torch.from_numpy(mix).cuda()
    Y = stft(x, 512, 128, layout='f t d', dtype=np.complex64)
    assert Y.is_cuda and Y.shape == (257, 128, D)
    gamma = CACGMMTrainer().fit_predict(Y, num_classes=K, iterations=10)
    assert gamma.is_cuda and gamma.shape == (257, K, 128)
    ref = stft(x[:1], 512, 128)[0]                                   # (T, F)
    est = istft(ref[None] * gamma.permute(1, 2, 0), 512, 128)[..., :N]
    assert est.is_cuda and est.shape == (K, N)
    np.testing.assert_allclose(est.sum(0).cpu().numpy(), mix[0], atol=1e-9)  # masks sum to one
