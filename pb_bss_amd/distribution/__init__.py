"""Drop-in for the hot-path part of pb_bss.distribution
(reference: pb_bss/distribution/__init__.py)."""
from .complex_angular_central_gaussian import (
    ComplexAngularCentralGaussian,
    ComplexAngularCentralGaussianTrainer,
    normalize_observation,
)
from .cacgmm import CACGMM, CACGMMTrainer
from .complex_watson import ComplexWatson, ComplexWatsonTrainer
from .cwmm import CWMM, CWMMTrainer

__all__ = [
    'CACGMM', 'CACGMMTrainer', 'CWMM', 'CWMMTrainer',
    'ComplexWatson', 'ComplexWatsonTrainer',
    'ComplexAngularCentralGaussian', 'ComplexAngularCentralGaussianTrainer',
    'normalize_observation',
]
