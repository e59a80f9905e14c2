"""Complex-Watson mixture model and EM trainer backed by the persistent HIP
kernel `cwmm_em_kernel` (csrc/cwmm.hpp).

Mirrors pb_bss/distribution/cwmm.py: `CWMM` (weight, complex_watson; predict)
and `CWMMTrainer` (fit / fit_predict) with the reference's arguments and
assertions.  Fused single-launch path for weight_constant_axis in
{(-1,), -1, -2} without an inline aligner; options that couple frequency bins
run E- and M-steps per iteration with the host hook in between (same entry
point with iterations = 0 / 1).
"""
from dataclasses import dataclass
from functools import cached_property
from operator import xor

import numpy as np

from .. import _lib, engine
from .cacgmm import CACGMMTrainer
from .complex_watson import ComplexWatson, ComplexWatsonTrainer
from .mixture_model_utils import (
    apply_inline_permutation_alignment,
    estimate_mixture_weight,
)
from .utils import _ProbabilisticModel, as_result

__all__ = ['CWMM', 'CWMMTrainer']


def _model_to_device(model, indep, K, D, device):
    t = _lib.torch()
    mode = _lib.to_device(model.complex_watson.mode, t.complex128).to(device)
    conc = _lib.to_device(model.complex_watson.concentration, t.float64).to(device)
    w = _lib.to_device(model.weight, t.float64).to(device)
    assert w.shape[-1] == 1, w.shape
    return (mode.expand(*indep, K, D).reshape(-1, K, D).contiguous(),
            conc.expand(*indep, K).reshape(-1, K).contiguous(),
            w.expand(*indep, K, 1).reshape(-1, K).contiguous())


@dataclass
class CWMM(_ProbabilisticModel):
    weight: np.ndarray = None  # (..., K, 1)
    complex_watson: ComplexWatson = None

    def predict(self, y):
        """y (..., N, D) -> affiliations (..., K, N) (reference :25-38; the
        observation is unit-normalised inside the kernel)."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert y.dtype in (t.complex64, t.complex128), y.dtype
        *indep, N, D = y.shape
        K = self.complex_watson.mode.shape[-2]
        dev_model = _model_to_device(self, tuple(indep), K, D, y.device)
        r = engine.cwmm_fit(y.reshape(-1, N, D).contiguous(), K, None, model=dev_model,
                            iterations=0, final_predict=True)
        return as_result(r['affiliation'].reshape(*indep, K, N), like_torch)

    _predict = predict  # the kernel normalises; already-normalised input is unchanged by it


class CWMMTrainer:
    def __init__(self, dimension=None, max_concentration=500, spline_markers=1000):
        self.dimension = dimension
        self.max_concentration = max_concentration
        self.spline_markers = spline_markers

    @cached_property
    def complex_watson_trainer(self):
        return ComplexWatsonTrainer(self.dimension, max_concentration=self.max_concentration,
                                    spline_markers=self.spline_markers)

    def fit(self, y, initialization=None, num_classes=None, iterations=100, *,
            saliency=None, weight_constant_axis=(-1,), affiliation_eps=0,
            inline_permutation_aligner=None):
        """EM for complex-Watson mixtures, any number of independent axes
        (reference :76-149).  y (..., T, D); initialization (..., K, T)."""
        assert xor(initialization is None, num_classes is None), (
            "Incompatible input combination. "
            "Exactly one of the two inputs has to be None: "
            f"{initialization is None} xor {num_classes is None}"
        )
        assert affiliation_eps == 0, affiliation_eps  # reference :161
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert y.dtype in (t.complex64, t.complex128), y.dtype
        assert y.shape[-1] > 1
        *indep, N, D = y.shape
        indep = tuple(indep)
        if initialization is None:
            shape = (*indep, num_classes, N)
            init = np.random.uniform(size=shape)  # global RNG, as the reference (:126-131)
            init /= np.einsum('...kn->...n', init)[..., None, :]
            gamma0 = _lib.to_device(init, t.float64).to(y.device)
        else:
            gamma0 = _lib.to_device(initialization, t.float64).to(y.device)
            num_classes = gamma0.shape[-2]
            gamma0 = gamma0.expand(*indep, num_classes, N)
        K = num_classes
        if self.dimension is None:
            self.dimension = D
        else:
            assert self.dimension == D, (
                'You initialized the trainer with a different dimension than '
                'you are using to fit a model. Use a new trainer, when you '
                'change the dimension.')
        if isinstance(weight_constant_axis, list):
            weight_constant_axis = tuple(weight_constant_axis)
        sal = None
        if saliency is not None:
            sal = _lib.to_device(saliency, t.float64).to(y.device).expand(*indep, N)
            sal = sal.reshape(-1, N).contiguous()
        yb = y.reshape(-1, N, D).contiguous()
        spline = self.complex_watson_trainer.device_spline(yb.device)
        mode = CACGMMTrainer._weight_mode(weight_constant_axis, len(indep) + 2)
        if mode is not None and inline_permutation_aligner is None:
            r = engine.cwmm_fit(yb, K, spline, gamma0=gamma0.reshape(-1, K, N).contiguous(),
                                iterations=iterations, saliency=sal, weight_mode=mode)
            if mode == _lib.WEIGHT_UNIFORM:
                weight = t.full((K, 1), 1.0 / K, dtype=t.float64, device=yb.device)
            else:
                weight = r['weight'].reshape(*indep, K, 1)
            return CWMM(
                weight=as_result(weight, like_torch),
                complex_watson=ComplexWatson(
                    mode=as_result(r['mode'].reshape(*indep, K, D), like_torch),
                    concentration=as_result(r['concentration'].reshape(*indep, K), like_torch)))
        return self._fit_stepwise(yb, indep, K, gamma0, iterations, saliency, sal,
                                  weight_constant_axis, inline_permutation_aligner, spline,
                                  like_torch)

    def _fit_stepwise(self, yb, indep, K, gamma0, iterations, saliency, sal,
                      weight_constant_axis, aligner, spline, like_torch):
        """The reference loop (:151-182) with device E/M steps and the host hook."""
        t = _lib.torch()
        B, N, D = yb.shape
        shape = (*indep, K, N)
        aff = _lib.to_host(gamma0.reshape(shape))
        sal_host = np.ones((*indep, N)) if saliency is None else np.broadcast_to(
            _lib.to_host(_lib.to_device(saliency, t.float64)), (*indep, N))
        model = None
        for _ in range(iterations):
            if model is not None:
                w = np.broadcast_to(model['weight'], (*indep, K, model['weight'].shape[-1]))
                if w.shape[-1] != 1:
                    raise NotImplementedError(
                        'frame-varying mixture weights (weight_constant_axis without -1) '
                        'are not supported by the Watson kernel')
                r = engine.cwmm_fit(
                    yb, K, None, model=(_lib.to_device(model['mode']),
                                        _lib.to_device(model['concentration']),
                                        _lib.to_device(np.ascontiguousarray(w[..., 0]).reshape(B, K))),
                    iterations=0, final_predict=True)
                aff = _lib.to_host(r['affiliation']).reshape(shape)
                if aligner is not None:
                    aff = apply_inline_permutation_alignment(
                        affiliation=aff, weight_constant_axis=weight_constant_axis,
                        aligner=aligner)
            weight = estimate_mixture_weight(aff, sal_host, weight_constant_axis)
            masked = aff * sal_host[..., None, :]
            r = engine.cwmm_fit(
                yb, K, spline,
                gamma0=_lib.to_device(np.ascontiguousarray(masked).reshape(B, K, N)),
                iterations=1)
            model = dict(weight=weight, mode=_lib.to_host(r['mode']),
                         concentration=_lib.to_host(r['concentration']))
        out = CWMM(weight=model['weight'],
                   complex_watson=ComplexWatson(
                       mode=model['mode'].reshape(*indep, K, D),
                       concentration=model['concentration'].reshape(*indep, K)))
        if like_torch:
            out = CWMM(weight=_lib.to_device(out.weight),
                       complex_watson=ComplexWatson(
                           mode=_lib.to_device(out.complex_watson.mode),
                           concentration=_lib.to_device(out.complex_watson.concentration)))
        return out

    def fit_predict(self, y, initialization=None, num_classes=None, iterations=100, *,
                    saliency=None, weight_constant_axis=(-1,), affiliation_eps=0,
                    inline_permutation_aligner=None):
        """Fit a model, then return the posterior affiliations (reference :184-210)."""
        model = self.fit(y=y, initialization=initialization, num_classes=num_classes,
                         iterations=iterations, saliency=saliency,
                         weight_constant_axis=weight_constant_axis,
                         affiliation_eps=affiliation_eps,
                         inline_permutation_aligner=inline_permutation_aligner)
        return model.predict(y)
