"""von-Mises-Fisher x complex-angular-central-Gaussian mixture model on the HIP
engine.  Mirrors pb_bss/distribution/vmfcacgmm.py:34-301: `VMFCACGMM` (predict)
and `VMFCACGMMTrainer` (fit / fit_predict).

As in the reference, the M-step sees the embedding as given (vmfcacgmm.py:267-276)
while the vMF log-pdf unit-normalises it (von_mises_fisher.py:71-73).
"""
from dataclasses import dataclass
from operator import xor
from typing import Any

from .. import _lib
from . import _joint
from .complex_angular_central_gaussian import ComplexAngularCentralGaussian
from .utils import _ProbabilisticModel, as_result
from .von_mises_fisher import VonMisesFisher
from ..utils import unsqueeze  # noqa: F401  (names the reference module exposes)
from .complex_angular_central_gaussian import ComplexAngularCentralGaussianTrainer  # noqa: F401  (names the reference module exposes)
from .von_mises_fisher import VonMisesFisherTrainer  # noqa: F401  (names the reference module exposes)
from .mixture_model_utils import (  # noqa: F401
    log_pdf_to_affiliation,
    log_pdf_to_affiliation_for_integration_models_with_inline_pa,
)

__all__ = ['VMFCACGMM', 'VMFCACGMMTrainer']


@dataclass
class VMFCACGMM(_ProbabilisticModel):
    weight: Any = None  # Shape (), (K,), (F, K), (K, T)
    weight_constant_axis: tuple = None
    vmf: VonMisesFisher = None
    cacg: ComplexAngularCentralGaussian = None
    spatial_weight: float = 1.
    spectral_weight: float = 1.

    def predict(self, observation, embedding):
        """observation (F, T, D) complex, embedding (F, T, E) real -> (F, K, T) (:43-55)."""
        return _joint.predict(_lib.EMBED_VMF, self, self.vmf.mean, self.vmf.concentration,
                              observation, embedding)


class VMFCACGMMTrainer:
    def fit(self, observation, embedding, initialization=None, num_classes=None, iterations=100,
            saliency=None, min_concentration=1e-10, max_concentration=500, hermitize=True,
            covariance_norm='eigenvalue', eigenvalue_floor=1e-10, affiliation_eps=1e-10,
            weight_constant_axis=(-1,), spatial_weight=1., spectral_weight=1.,
            inline_permutation_alignment=False) -> VMFCACGMM:
        """(:101-205)."""
        assert xor(initialization is None, num_classes is None), (
            "Incompatible input combination. "
            "Exactly one of the two inputs has to be None: "
            f"{initialization is None} xor {num_classes is None}"
        )
        r, like_torch = _joint.fit(
            _lib.EMBED_VMF, observation, embedding, initialization, num_classes, iterations,
            saliency, covariance_norm=covariance_norm, eigenvalue_floor=eigenvalue_floor,
            affiliation_eps=affiliation_eps, weight_constant_axis=weight_constant_axis,
            spatial_weight=spatial_weight, spectral_weight=spectral_weight,
            inline_permutation_alignment=inline_permutation_alignment,
            min_concentration=min_concentration, max_concentration=max_concentration)
        mode = _joint.weight_mode(weight_constant_axis)
        K = r['mean'].shape[0]
        return VMFCACGMM(
            weight=_joint.weight_of(r, mode, K, like_torch),
            weight_constant_axis=tuple(weight_constant_axis) if not isinstance(
                weight_constant_axis, int) else (weight_constant_axis,),
            vmf=VonMisesFisher(mean=as_result(r['mean'], like_torch),
                               concentration=as_result(r['scale'], like_torch)),
            cacg=_joint.cacg_of(r, like_torch),
            spatial_weight=spatial_weight, spectral_weight=spectral_weight)

    def fit_predict(self, observation, embedding, initialization=None, num_classes=None,
                    iterations=100, saliency=None, min_concentration=1e-10,
                    max_concentration=500, hermitize=True, covariance_norm='eigenvalue',
                    eigenvalue_floor=1e-10, affiliation_eps=1e-10, weight_constant_axis=(-1,),
                    spatial_weight=1., spectral_weight=1., inline_permutation_alignment=False):
        """Fit a model. Then just return the posterior affiliations (:207-235)."""
        model = self.fit(
            observation=observation, embedding=embedding, initialization=initialization,
            num_classes=num_classes, iterations=iterations, saliency=saliency,
            min_concentration=min_concentration, max_concentration=max_concentration,
            hermitize=hermitize, covariance_norm=covariance_norm,
            eigenvalue_floor=eigenvalue_floor, affiliation_eps=affiliation_eps,
            weight_constant_axis=weight_constant_axis, spatial_weight=spatial_weight,
            spectral_weight=spectral_weight,
            inline_permutation_alignment=inline_permutation_alignment)
        return model.predict(observation=observation, embedding=embedding)
