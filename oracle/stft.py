"""CPU oracle of the STFT / inverse STFT edge ("STFT edge").

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

The reference has no STFT of its own: its tests and transform module call
``nara_wpe.utils.stft`` / ``istft`` (``tests/test_distribution/test_spatial_mm.py:4,17-22``,
``pb_bss/transform/griffin_lim_module.py:37``; ``setup.py:53`` lists ``nara_wpe`` unpinned, and
the package is absent from this container).  This file restates the published algorithm of
those two functions in plain NumPy: zero "fading" pads of ``window_length - shift`` samples on
both sides, a periodic window (``window(window_length + 1)[:-1]``), frames every ``shift``
samples with the last one zero-padded (``pad=True``) or dropped, ``numpy.fft.rfft(n=size)``;
the inverse multiplies ``irfft`` frames by the biorthogonal synthesis window
``w / sum_m w[n + m shift]^2`` and overlap-adds.  PARITY UNPINNED against nara_wpe itself (it
cannot be imported here, and the reference holds no STFT vector of its own); the restatement is
pinned instead (i) by two independent third-party implementations with matched window, padding
and hop -- scipy.signal.stft and torch.stft, tests/test_oracle_golden.py::
test_stft_oracle_against_scipy_and_torch --, (ii) by numpy.fft as ground truth for every frame, by
perfect reconstruction istft(stft(x)) == x and by a known-answer sinusoid
(::test_stft_oracle_properties).
"""
import numpy as np


def periodic_window(name_or_callable, window_length, symmetric_window=False):
    """``window(window_length + 1)[:-1]`` (periodic) or ``window(window_length)`` (symmetric)."""
    if callable(name_or_callable):
        fn = name_or_callable
    else:
        m = {'blackman': np.blackman, 'hann': np.hanning, 'hanning': np.hanning,
             'hamming': np.hamming}
        fn = m[name_or_callable]
    if symmetric_window:
        return np.asarray(fn(window_length), dtype=np.float64)
    return np.asarray(fn(window_length + 1), dtype=np.float64)[:-1]


def num_frames(num_samples, size, shift, window_length=None, fading=True, pad=True):
    wl = size if window_length is None else window_length
    n = num_samples + (2 * (wl - shift) if fading else 0)
    if n < wl:
        return 1 if pad else 0
    if pad:
        return 1 + -(-(n - wl) // shift)
    return 1 + (n - wl) // shift


def stft(x, size=1024, shift=256, window='blackman', window_length=None, fading=True, pad=True,
         symmetric_window=False):
    """(..., N) real -> (..., T, size // 2 + 1) complex128."""
    x = np.asarray(x, dtype=np.float64)
    wl = size if window_length is None else window_length
    if fading:
        pw = [(0, 0)] * (x.ndim - 1) + [(wl - shift, wl - shift)]
        x = np.pad(x, pw)
    w = periodic_window(window, wl, symmetric_window)
    T = num_frames(x.shape[-1], size, shift, wl, fading=False, pad=pad)
    need = (T - 1) * shift + wl
    if need > x.shape[-1]:
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, need - x.shape[-1])])
    frames = np.stack([x[..., t * shift:t * shift + wl] for t in range(T)], axis=-2)
    return np.fft.rfft(frames * w, n=size, axis=-1)


def biorthogonal_window(analysis_window, shift):
    wl = len(analysis_window)
    assert wl % shift == 0, (wl, shift)
    s = np.zeros(shift)
    for m in range(wl // shift):
        s += analysis_window[m * shift:(m + 1) * shift] ** 2
    return analysis_window / np.tile(s, wl // shift)


def istft(X, size=1024, shift=256, window='blackman', fading=True, window_length=None,
          symmetric_window=False):
    """(..., T, size // 2 + 1) complex -> (..., T * shift + window_length - shift [- 2 fade]) real."""
    X = np.asarray(X)
    assert X.shape[-1] == size // 2 + 1, X.shape
    wl = size if window_length is None else window_length
    w = biorthogonal_window(periodic_window(window, wl, symmetric_window), shift)
    T = X.shape[-2]
    out = np.zeros(X.shape[:-2] + (T * shift + wl - shift,))
    frames = np.fft.irfft(X, n=size, axis=-1)[..., :wl] * w
    for t in range(T):
        out[..., t * shift:t * shift + wl] += frames[..., t, :]
    if fading:
        out = out[..., wl - shift:out.shape[-1] - (wl - shift)]
    return out
