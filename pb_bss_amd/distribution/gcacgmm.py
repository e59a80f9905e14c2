"""Gaussian x complex-angular-central-Gaussian mixture model (Deep-Clustering
embeddings + spatial observations, Drude 2019) on the HIP engine.

Mirrors pb_bss/distribution/gcacgmm.py:38-333: `GCACGMM` (predict) and
`GCACGMMTrainer` (fit / fit_predict), no independent axes; covariance_type 'spherical' (the
default, csrc/embed.hip), 'diagonal' (as the reference writes its log-pdf) and 'full'
(csrc/gauss_full.hip on the FP64 matrix pipe).
"""
from dataclasses import dataclass
from operator import xor
from typing import Any

import numpy as np

from .. import _lib
from . import _joint
from .complex_angular_central_gaussian import ComplexAngularCentralGaussian
from .gaussian import DiagonalGaussian, Gaussian, SphericalGaussian
from .utils import _ProbabilisticModel, as_result
from ..utils import unsqueeze  # noqa: F401  (names the reference module exposes)
from .complex_angular_central_gaussian import ComplexAngularCentralGaussianTrainer  # noqa: F401  (names the reference module exposes)
from .gaussian import GaussianTrainer  # noqa: F401  (names the reference module exposes)
from .mixture_model_utils import (  # noqa: F401
    log_pdf_to_affiliation,
    log_pdf_to_affiliation_for_integration_models_with_inline_pa,
)

_KIND = {'spherical': _lib.EMBED_GAUSS_SPHERICAL, 'diagonal': _lib.EMBED_GAUSS_DIAG,
         'full': _lib.EMBED_GAUSS_FULL}
_CLASS = {'spherical': SphericalGaussian, 'diagonal': DiagonalGaussian, 'full': Gaussian}


def _kind_of(gaussian):
    for name, cls in _CLASS.items():
        if isinstance(gaussian, cls):
            return _KIND[name]
    raise TypeError(type(gaussian))


__all__ = ['GCACGMM', 'GCACGMMTrainer']


@dataclass
class GCACGMM(_ProbabilisticModel):
    weight: Any = None  # Shape (), (K,), (F, K), (K, T)
    weight_constant_axis: tuple = None
    gaussian: Any = None  # Gaussian, DiagonalGaussian, or SphericalGaussian
    cacg: ComplexAngularCentralGaussian = None
    spatial_weight: float = 1.
    spectral_weight: float = 1.

    def predict(self, observation, embedding):
        """observation (F, T, D) complex, embedding (F, T, E) real -> affiliation (F, K, T)
        (:47-64)."""
        return _joint.predict(_kind_of(self.gaussian), self, self.gaussian.mean,
                              self.gaussian.covariance, observation, embedding)


class GCACGMMTrainer:
    def fit(self, observation, embedding, initialization=None, num_classes=None, iterations=100,
            saliency=None, hermitize=True, covariance_norm='eigenvalue', eigenvalue_floor=1e-10,
            covariance_type='spherical', fixed_covariance=None, affiliation_eps=1e-10,
            weight_constant_axis=(-1,), spatial_weight=1., spectral_weight=1.,
            inline_permutation_alignment=False) -> GCACGMM:
        """(:131-246).  initialization (F, K, T); saliency (F, T)."""
        assert xor(initialization is None, num_classes is None), (
            "Incompatible input combination. "
            "Exactly one of the two inputs has to be None: "
            f"{initialization is None} xor {num_classes is None}"
        )
        if covariance_type not in _KIND:
            raise ValueError(f"Unknown covariance type '{covariance_type}'.")  # gaussian.py:184
        r, like_torch = _joint.fit(
            _KIND[covariance_type], observation, embedding, initialization, num_classes,
            iterations, saliency, covariance_norm=covariance_norm,
            eigenvalue_floor=eigenvalue_floor, affiliation_eps=affiliation_eps,
            weight_constant_axis=weight_constant_axis, spatial_weight=spatial_weight,
            spectral_weight=spectral_weight,
            inline_permutation_alignment=inline_permutation_alignment,
            fixed_scale=fixed_covariance)
        mode = _joint.weight_mode(weight_constant_axis)
        K = r['mean'].shape[0]
        return GCACGMM(
            weight=_joint.weight_of(r, mode, K, like_torch),
            weight_constant_axis=tuple(weight_constant_axis) if not isinstance(
                weight_constant_axis, int) else (weight_constant_axis,),
            gaussian=_CLASS[covariance_type](mean=as_result(r['mean'], like_torch),
                                             covariance=as_result(r['scale'], like_torch)),
            cacg=_joint.cacg_of(r, like_torch),
            spatial_weight=spatial_weight, spectral_weight=spectral_weight)

    def fit_predict(self, observation, embedding, initialization=None, num_classes=None,
                    iterations=100, saliency=None, hermitize=True, covariance_norm='eigenvalue',
                    eigenvalue_floor=1e-10, covariance_type='spherical', fixed_covariance=None,
                    affiliation_eps=1e-10, weight_constant_axis=(-1,), spatial_weight=1.,
                    spectral_weight=1., inline_permutation_alignment=False):
        """Fit a model. Then just return the posterior affiliations (:248-265)."""
        model = self.fit(
            observation=observation, embedding=embedding, initialization=initialization,
            num_classes=num_classes, iterations=iterations, saliency=saliency,
            hermitize=hermitize, covariance_norm=covariance_norm,
            eigenvalue_floor=eigenvalue_floor, covariance_type=covariance_type,
            fixed_covariance=fixed_covariance, affiliation_eps=affiliation_eps,
            weight_constant_axis=weight_constant_axis, spatial_weight=spatial_weight,
            spectral_weight=spectral_weight,
            inline_permutation_alignment=inline_permutation_alignment)
        return model.predict(observation=observation, embedding=embedding)
