"""Drop-in for the hot-path part of pb_bss.distribution
(reference: pb_bss/distribution/__init__.py)."""
from .complex_angular_central_gaussian import (
    ComplexAngularCentralGaussian,
    ComplexAngularCentralGaussianTrainer,
    normalize_observation,
    sample_complex_angular_central_gaussian,
)
from .cacgmm import CACGMM, CACGMMTrainer, sample_cacgmm
from .complex_watson import ComplexWatson, ComplexWatsonTrainer
from .cwmm import CWMM, CWMMTrainer
from .von_mises_fisher import VonMisesFisher, VonMisesFisherTrainer
from .vmfmm import VMFMM, VMFMMTrainer
from .gaussian import DiagonalGaussian, Gaussian, SphericalGaussian, GaussianTrainer
from .gmm import GMM, GMMTrainer
from .gcacgmm import GCACGMM, GCACGMMTrainer
from .vmfcacgmm import VMFCACGMM, VMFCACGMMTrainer

__all__ = [
    'CACGMM', 'CACGMMTrainer', 'CWMM', 'CWMMTrainer',
    'VonMisesFisher', 'VonMisesFisherTrainer', 'VMFMM', 'VMFMMTrainer',
    'Gaussian', 'DiagonalGaussian', 'SphericalGaussian', 'GaussianTrainer', 'GMM', 'GMMTrainer', 'GCACGMM', 'GCACGMMTrainer',
    'VMFCACGMM', 'VMFCACGMMTrainer',
    'ComplexWatson', 'ComplexWatsonTrainer',
    'ComplexAngularCentralGaussian', 'ComplexAngularCentralGaussianTrainer',
    'normalize_observation', 'sample_cacgmm', 'sample_complex_angular_central_gaussian',
]
