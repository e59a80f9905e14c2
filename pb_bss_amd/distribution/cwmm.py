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
from .mixture_model_utils import (  # noqa: F401  (re-exported like the reference's cwmm.py)
    apply_inline_permutation_alignment,
    estimate_mixture_weight,
    log_pdf_to_affiliation,
)
from .complex_angular_central_gaussian import normalize_observation  # noqa: F401
from .utils import _ProbabilisticModel, as_result, random_affiliation

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
        w = _lib.to_device(self.weight, t.float64).to(y.device)
        if w.shape[-1] != 1:
            # frame-varying weights (weight_constant_axis without -1, reference :40-52 with a
            # (..., K, N) weight): class log-pdfs, then the general softmax step
            yb = y.reshape(-1, N, D).contiguous()
            B = yb.shape[0]
            mode = _lib.to_device(self.complex_watson.mode, t.complex128).to(y.device)
            conc = _lib.to_device(self.complex_watson.concentration, t.float64).to(y.device)
            r = engine.cwmm_fit(
                yb, K, None,
                model=(mode.expand(*indep, K, D).reshape(B, K, D).contiguous(),
                       conc.expand(*indep, K).reshape(B, K).contiguous(),
                       t.ones((B, K), dtype=t.float64, device=y.device)),
                iterations=0, want_log_pdf=True)
            while w.ndim < len(indep) + 2:
                w = w.unsqueeze(0)
            w = (w.expand(*indep, *w.shape[-2:]).reshape(-1, *w.shape[-2:])
                 if any(a != 1 for a in w.shape[:-2]) else w.reshape(1, *w.shape[-2:]))
            aff = engine.log_pdf_to_affiliation(r['log_pdf'], w)
            return as_result(aff.reshape(*indep, K, N), like_torch)
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
            gamma0 = random_affiliation(shape, y.device)  # global NumPy RNG (:126-131)
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
        # weights averaged over the bins of an utterance ((-3, -1), cwmm.py:217-240): one
        # cooperative launch (csrc/cwmm.hpp: WatsonShared); None: not served / timed out
        axes = {a % (len(indep) + 2) - (len(indep) + 2) for a in (
            (weight_constant_axis,) if isinstance(weight_constant_axis, int)
            else weight_constant_axis)}
        if axes in ({-3, -1}, {-3}) and len(indep) >= 1 and inline_permutation_aligner is None:
            group = indep[-1]
            kt = axes == {-3}
            r = engine.cwmm_fit(yb, K, spline, gamma0=gamma0.reshape(-1, K, N).contiguous(),
                                iterations=iterations, saliency=sal, group=group,
                                weight_mode=_lib.WEIGHT_SHARED_KT if kt else _lib.WEIGHT_SHARED_K)
            if r is not None:
                weight = r['weight'].reshape(*indep[:-1], 1, K, N if kt else 1)
                return CWMM(
                    weight=as_result(weight, like_torch),
                    complex_watson=ComplexWatson(
                        mode=as_result(r['mode'].reshape(*indep, K, D), like_torch),
                        concentration=as_result(r['concentration'].reshape(*indep, K),
                                                like_torch)))
        return self._fit_stepwise(yb, indep, K, gamma0, iterations, saliency, sal,
                                  weight_constant_axis, inline_permutation_aligner, spline,
                                  like_torch)

    def _fit_stepwise(self, yb, indep, K, gamma0, iterations, saliency, sal,
                      weight_constant_axis, aligner, spline, like_torch):
        """The reference loop (:151-182) for the options that couple the bins
        (`weight_constant_axis` with independent axes, frame-varying weights, an inline aligner),
        every step a device kernel: class log-pdfs (`pbbss_cwmm_fit`, iterations = 0), the
        softmax with the reference-shaped weight (`pbbss_log_pdf_to_affiliation`), the weight
        reduction (`pbbss_estimate_mixture_weight`) and the M-step (`pbbss_cwmm_fit`,
        iterations = 1).  Nothing returns to the host inside the loop, except for a foreign
        (NumPy) aligner object."""
        from . import _embed_stepwise as sw
        t = _lib.torch()
        B, N, D = yb.shape
        shape = (*indep, K, N)
        aff = gamma0.reshape(shape).contiguous()
        sal_dev = None if sal is None else sal.reshape(*indep, N)
        ones_w = t.ones((B, K), dtype=t.float64, device=yb.device)
        mode = conc = weight = None
        for _ in range(iterations):
            if mode is not None:
                r = engine.cwmm_fit(yb, K, None, model=(mode, conc, ones_w), iterations=0,
                                    want_log_pdf=True)
                w = weight
                while w.ndim < len(shape):
                    w = w.unsqueeze(0)
                w = (w.expand(*indep, *w.shape[-2:]).reshape(-1, *w.shape[-2:])
                     if any(a != 1 for a in w.shape[:-2]) else w.reshape(1, *w.shape[-2:]))
                aff = engine.log_pdf_to_affiliation(r['log_pdf'], w).reshape(shape)
                if aligner is not None:
                    if type(aligner).__module__.startswith('pb_bss_amd'):
                        aff = apply_inline_permutation_alignment(
                            affiliation=aff, weight_constant_axis=weight_constant_axis,
                            aligner=aligner).contiguous()
                    else:  # a foreign (NumPy) aligner object: the one host excursion left
                        aff = _lib.to_device(apply_inline_permutation_alignment(
                            affiliation=_lib.to_host(aff),
                            weight_constant_axis=weight_constant_axis, aligner=aligner),
                            t.float64).to(yb.device).contiguous()
            weight = sw.device_weight(aff, sal_dev, weight_constant_axis, indep)
            masked = aff if sal_dev is None else aff * sal_dev[..., None, :]
            r = engine.cwmm_fit(yb, K, spline, gamma0=masked.reshape(B, K, N).contiguous(),
                                iterations=1)
            mode, conc = r['mode'], r['concentration']
        return CWMM(weight=as_result(weight, like_torch),
                    complex_watson=ComplexWatson(
                        mode=as_result(mode.reshape(*indep, K, D), like_torch),
                        concentration=as_result(conc.reshape(*indep, K), like_torch)))

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
