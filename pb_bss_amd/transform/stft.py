"""`stft` / `istft` with the signatures of ``nara_wpe.utils.stft`` / ``istft``, the functions
the reference's tests and ``pb_bss/transform/griffin_lim_module.py:37`` call
(``tests/test_distribution/test_spatial_mm.py:4,17-22``: ``_stft(signal, 512, 128)`` and
``_istft(signal, 512, 128)[..., :num_samples]``).  The transforms run in
``pb_bss_amd/csrc/stft.hip`` (LDS Stockham FFT); this module only prepares the windows (a few
hundred numbers, NumPy) and the shapes.  NumPy in -> NumPy out, torch CUDA in -> torch CUDA out.

Extension over the reference call: ``layout='f t d'`` writes the (bins, frames, channels)
arrangement that every mixture-model trainer is fed with (``rearrange(Y, 'd t f -> f t d')``,
test_spatial_mm.py:41) directly from the kernel.
"""
import numpy as np

from .. import _lib, engine

_WINDOWS = {'blackman': np.blackman, 'hann': np.hanning, 'hanning': np.hanning,
            'hamming': np.hamming}


def analysis_window(window, window_length, symmetric_window=False):
    """Periodic (``window(window_length + 1)[:-1]``) or symmetric window as float64."""
    fn = _WINDOWS[window] if isinstance(window, str) else window
    if symmetric_window:
        w = fn(window_length)
    else:
        w = fn(window_length + 1)[:-1]
    w = np.ascontiguousarray(w, dtype=np.float64)
    assert w.shape == (window_length,), (w.shape, window_length)
    return w


def biorthogonal_window(analysis, shift):
    """Synthesis window ``w / sum_m w[n + m shift]^2`` (perfect reconstruction with
    overlap-add for ``window_length % shift == 0``)."""
    analysis = np.asarray(analysis, dtype=np.float64)
    wl = analysis.shape[0]
    if wl % shift:
        raise ValueError(f'window_length {wl} must be a multiple of shift {shift}')
    s = (analysis.reshape(wl // shift, shift) ** 2).sum(axis=0)
    return np.ascontiguousarray(analysis / np.tile(s, wl // shift))


def stft_frames_to_samples(frames, size, shift, window_length=None, fading=True):
    """Length of the signal ``istft`` returns for ``frames`` frames."""
    wl = size if window_length is None else window_length
    return frames * shift + wl - shift - (2 * (wl - shift) if fading else 0)


def stft(time_signal, size=1024, shift=256, axis=-1, window='blackman', window_length=None,
         fading=True, pad=True, symmetric_window=False, *, layout=None, dtype=None):
    """Short-time Fourier transform along ``axis``.

    time_signal (..., samples) real -> (..., frames, size // 2 + 1) complex128 (the frame axis
    takes the place of ``axis``, the frequency axis follows it).  ``layout='f t d'`` (2-D input
    (channels, samples) only): (bins, frames, channels) instead.  ``dtype``: numpy complex64 /
    complex128 of the result (default complex128, what numpy.fft.rfft returns).
    """
    t = _lib.torch()
    like_torch = _lib.is_torch(time_signal)
    wl = size if window_length is None else window_length
    x = _lib.to_device(time_signal)
    if x.dtype not in (t.float32, t.float64):
        x = x.to(t.float64)
    nd = x.ndim
    ax = axis % nd
    if ax != nd - 1:
        x = x.movedim(ax, -1)
    lead = tuple(x.shape[:-1])
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    w = _lib.to_device(analysis_window(window, wl, symmetric_window), t.float64, device=x2.device)
    c128 = dtype is None or np.dtype(dtype) == np.complex128
    if layout is not None:
        if layout.replace(' ', '') != 'ftd':
            raise ValueError(f"layout must be None or 'f t d', got {layout!r}")
        if nd != 2 or ax != 1:
            raise ValueError("layout='f t d' needs a (channels, samples) signal")
        out = engine.stft(x2, size, shift, w, fading=fading, pad=pad, layout=1, out_c128=c128)
        return out if like_torch else _lib.to_host(out)
    out = engine.stft(x2, size, shift, w, fading=fading, pad=pad, layout=0, out_c128=c128)
    out = out.reshape(lead + tuple(out.shape[-2:]))
    if ax != nd - 1:
        out = out.movedim((-2, -1), (ax, ax + 1))
    return out if like_torch else _lib.to_host(out)


def istft(stft_signal, size=1024, shift=256, window='blackman', fading=True, window_length=None,
          symmetric_window=False):
    """Inverse of `stft`: (..., frames, size // 2 + 1) complex -> (..., samples) float64 with
    samples = `stft_frames_to_samples` (callers cut to the original length themselves, as the
    reference's tests do)."""
    t = _lib.torch()
    like_torch = _lib.is_torch(stft_signal)
    wl = size if window_length is None else window_length
    X = _lib.to_device(stft_signal)
    if X.dtype not in (t.complex64, t.complex128):
        X = X.to(t.complex128)
    if X.shape[-1] != size // 2 + 1:
        raise AssertionError(f'last axis {X.shape[-1]} != size // 2 + 1 = {size // 2 + 1}')
    lead = tuple(X.shape[:-2])
    X3 = X.reshape((-1,) + tuple(X.shape[-2:])).contiguous()
    w = biorthogonal_window(analysis_window(window, wl, symmetric_window), shift)
    wd = _lib.to_device(w, t.float64, device=X3.device)
    out = engine.istft(X3, size, shift, wd, fading=fading)
    out = out.reshape(lead + (out.shape[-1],))
    return out if like_torch else _lib.to_host(out)
