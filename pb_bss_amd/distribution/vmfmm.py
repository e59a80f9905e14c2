"""von Mises-Fisher mixture model (clustering of unit-norm embeddings) on the HIP
embedding kernels.  Mirrors pb_bss/distribution/vmfmm.py:14-172: `VMFMM`
(vmf, weight; predict) and `VMFMMTrainer` (fit / fit_predict).

The whole EM loop is one C-ABI call (`pbbss_vmfmm_fit`): per iteration one
E-step kernel over the transposed embedding and one M-step reduction over the
row-major embedding, enqueued back to back on the caller's stream.
"""
from dataclasses import dataclass
from operator import xor

import numpy as np

from .. import _lib, engine
from .utils import _ProbabilisticModel, as_result, random_affiliation
from .von_mises_fisher import VonMisesFisher
from .von_mises_fisher import VonMisesFisherTrainer  # noqa: F401  (names the reference module exposes)
from .mixture_model_utils import estimate_mixture_weight, log_pdf_to_affiliation  # noqa: F401  (names the reference module exposes)

__all__ = ['VMFMM', 'VMFMMTrainer']


def _weight_mode(weight_constant_axis, ndim):
    """(-1,) / -1 -> per-mixture class weights; int -2 -> uniform 1/K (the only
    spelling the reference's estimate_mixture_weight maps to 1/K)."""
    if isinstance(weight_constant_axis, list):
        weight_constant_axis = tuple(weight_constant_axis)
    if isinstance(weight_constant_axis, int):
        if weight_constant_axis % ndim - ndim == -2:
            return _lib.WEIGHT_UNIFORM
        weight_constant_axis = (weight_constant_axis,)
    if tuple(a % ndim - ndim for a in weight_constant_axis) == (-1,):
        return _lib.WEIGHT_PER_CLASS_MEAN
    return None  # any other axis set: the step-wise device loop (_embed_stepwise.py)


@dataclass
class VMFMM(_ProbabilisticModel):
    vmf: VonMisesFisher = None
    weight: np.ndarray = None  # (..., K, 1)

    def predict(self, y):
        """y (..., N, D) real -> affiliations (..., K, N) (:19-31)."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        *indep, N, E = y.shape
        mean = _lib.to_device(self.vmf.mean, t.float64).to(y.device)
        K = mean.shape[-2]
        conc = _lib.to_device(self.vmf.concentration, t.float64).to(y.device)
        w = _lib.to_device(self.weight, t.float64).to(y.device)
        if w.shape[-1] != 1 or (w.ndim > 2 and tuple(w.shape[:-2]) != tuple(indep)
                                and any(a != 1 for a in w.shape[:-2])):
            # frame-varying weights (weight_constant_axis without -1): the general softmax step
            from . import _embed_stepwise as sw
            aff = sw.affiliation(
                'vmf', y.reshape(-1, N, E), mean.expand(*indep, K, E).reshape(-1, K, E).contiguous(),
                conc.expand(*indep, K).reshape(-1, K).contiguous(), w, tuple(indep), K, N)
            return as_result(aff.reshape(*indep, K, N), like_torch)
        model = (mean.expand(*indep, K, E).reshape(-1, K, E).contiguous(),
                 conc.expand(*indep, K).reshape(-1, K).contiguous(),
                 w.expand(*indep, K, 1).reshape(-1, K).contiguous())
        r = engine.vmfmm_fit(y.reshape(-1, N, E), K, model=model, iterations=0,
                             final_predict=True)
        return as_result(r['affiliation'].reshape(*indep, K, N), like_torch)

    _predict = predict  # normalising unit rows again is the identity


class VMFMMTrainer:
    """The vMFMM can be used to cluster the embeddings."""

    def fit(self, y, initialization=None, num_classes=None, iterations=100, saliency=None,
            weight_constant_axis=(-1,), min_concentration=1e-10, max_concentration=500):
        """EM for vMFMMs with any number of independent dimensions (:42-104).
        y (..., N, D) real; initialization (..., K, N); saliency (..., N)."""
        assert xor(initialization is None, num_classes is None), (
            "Incompatible input combination. "
            "Exactly one of the two inputs has to be None: "
            f"{initialization is None} xor {num_classes is None}"
        )
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        *indep, N, E = y.shape
        indep = tuple(indep)
        if initialization is None:
            # global NumPy RNG (:83-86)
            gamma0 = random_affiliation((*indep, num_classes, N), y.device)
        else:
            gamma0 = _lib.to_device(initialization, t.float64).to(y.device)
            num_classes = gamma0.shape[-2]
            gamma0 = gamma0.expand(*indep, num_classes, N)
        K = num_classes
        assert iterations > 0, iterations
        mode = _weight_mode(weight_constant_axis, len(indep) + 2)
        if mode is None:
            from . import _embed_stepwise as sw
            r = sw.fit('vmf', y, gamma0.contiguous(), iterations, saliency, weight_constant_axis,
                       min_concentration=min_concentration, max_concentration=max_concentration)
            return VMFMM(
                weight=as_result(r['weight'], like_torch),
                vmf=VonMisesFisher(
                    mean=as_result(r['mean'].reshape(*indep, K, E), like_torch),
                    concentration=as_result(r['scale'].reshape(*indep, K), like_torch)))
        sal = None
        if saliency is not None:
            sal = _lib.to_device(saliency, t.float64).to(y.device).expand(*indep, N)
            sal = sal.reshape(-1, N).contiguous()
        r = engine.vmfmm_fit(y.reshape(-1, N, E), K, gamma0=gamma0.reshape(-1, K, N).contiguous(),
                             iterations=iterations, saliency=sal, weight_mode=mode,
                             min_concentration=min_concentration,
                             max_concentration=max_concentration)
        if mode == _lib.WEIGHT_UNIFORM:
            weight = t.full((K, 1), 1.0 / K, dtype=t.float64, device=y.device)
        else:
            weight = r['weight'].reshape(*indep, K, 1)
        return VMFMM(
            weight=as_result(weight, like_torch),
            vmf=VonMisesFisher(
                mean=as_result(r['mean'].reshape(*indep, K, E), like_torch),
                concentration=as_result(r['concentration'].reshape(*indep, K), like_torch)))

    def fit_predict(self, y, initialization=None, num_classes=None, iterations=100,
                    saliency=None, weight_constant_axis=(-1,), min_concentration=1e-10,
                    max_concentration=500):
        """Fit a model. Then just return the posterior affiliations (:106-127)."""
        model = self.fit(y=y, initialization=initialization, num_classes=num_classes,
                         iterations=iterations, saliency=saliency,
                         min_concentration=min_concentration,
                         max_concentration=max_concentration,
                         weight_constant_axis=weight_constant_axis)
        return model.predict(y)
