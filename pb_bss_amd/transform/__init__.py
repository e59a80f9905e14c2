"""STFT edge of the separation path: audio -> STFT -> (mixture model, beamformer) -> audio
without leaving the device."""
from .stft import stft, istft, stft_frames_to_samples, biorthogonal_window, analysis_window

__all__ = ['stft', 'istft', 'stft_frames_to_samples', 'biorthogonal_window', 'analysis_window']
