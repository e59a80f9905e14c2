"""Complex Watson distribution on the device.

Mirrors pb_bss/distribution/complex_watson.py: `ComplexWatson` (mode,
concentration; log_pdf / log_norm) and `ComplexWatsonTrainer` (the
concentration look-up: a quadratic spline of the inverse hypergeometric
ratio, built on the host with SciPy exactly as the reference does, then
evaluated on the device with de Boor's recurrence inside the EM kernel).
"""
import math
from dataclasses import dataclass
from functools import cached_property

import numpy as np

from .. import _lib, engine
from .utils import _ProbabilisticModel, as_result
from ..utils import get_pca, is_broadcast_compatible  # noqa: F401  (names the reference module exposes)

__all__ = ['ComplexWatson', 'ComplexWatsonTrainer', 'normalize_observation']


def normalize_observation(observation):
    """(..., N, D) / max(norm, tiny); layout unchanged (reference :16-29).
    Host/torch elementwise helper -- the EM kernel normalises on load itself."""
    if _lib.is_torch(observation):
        t = _lib.torch()
        n = t.linalg.vector_norm(observation, dim=-1, keepdim=True)
        return observation / t.clamp(n, min=np.finfo(np.float64).tiny)
    return observation / np.maximum(
        np.linalg.norm(observation, axis=-1, keepdims=True),
        np.finfo(observation.dtype).tiny)


@dataclass
class ComplexWatson(_ProbabilisticModel):
    mode: np.ndarray = None  # (..., D)
    concentration: np.ndarray = None  # (...)

    @staticmethod
    def log_norm_1f1(scale, dimension):
        """ln(1F1(1; D; kappa) * 2 pi^D / (D-1)!)  (reference :157-168).
        Host scalar helper (SciPy), used for inspection; the kernel has its
        own closed form (csrc/cwmm.hpp: watson_log_norm)."""
        from scipy.special import hyp1f1
        norm = hyp1f1(1, dimension, scale) * (
            2 * np.pi ** dimension / math.factorial(dimension - 1))
        return np.log(norm)

    def log_norm(self):
        conc = self.concentration
        if _lib.is_torch(conc):
            conc = _lib.to_host(conc)
        return self.log_norm_1f1(conc, self.mode.shape[-1])

    def pdf(self, y):
        """exp(log_pdf(y)) (reference :61-71)."""
        lp = self.log_pdf(y)
        return lp.exp() if _lib.is_torch(lp) else np.exp(lp)

    def log_pdf(self, y):
        """y (..., N, D) unit norm -> (..., N) after broadcasting with the
        parameter axes; mixture models pass y[..., None, :, :] against
        mode (..., K, D) (reference :73-87)."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        if y.dtype not in (t.complex64, t.complex128):
            y = y.to(t.complex128)
        mode = _lib.to_device(self.mode, t.complex128)
        conc = _lib.to_device(self.concentration, t.float64)
        *y_indep, N, D = y.shape
        p_indep = tuple(mode.shape[:-1])
        y_indep = (1,) * (len(p_indep) - len(y_indep)) + tuple(y_indep)
        full = tuple(np.broadcast_shapes(y_indep, p_indep))
        if len(full) >= 1 and y_indep[-1] == 1:
            lead, K = full[:-1], full[-1]
            yb = y.reshape(*y_indep[:-1], N, D)
        else:
            lead, K = full, 1
            yb = y.reshape(*y_indep, N, D)
        B = int(np.prod(lead)) if lead else 1
        yb = yb.expand(*lead, N, D).reshape(B, N, D).contiguous()
        if K == 1 and full == lead:
            mb = mode.expand(*lead, D).reshape(B, 1, D).contiguous()
            cb = conc.expand(*lead).reshape(B, 1).contiguous()
        else:
            mb = mode.expand(*lead, K, D).reshape(B, K, D).contiguous()
            cb = conc.expand(*lead, K).reshape(B, K).contiguous()
        w = t.ones((B, K), dtype=t.float64, device=yb.device)
        r = engine.cwmm_fit(yb, K, None, model=(mb, cb, w), iterations=0, want_log_pdf=True)
        lp = r['log_pdf']
        lp = lp.reshape(*full, N) if (K == 1 and full == lead) else lp.reshape(*lead, K, N)
        return as_result(lp, like_torch)


class ComplexWatsonTrainer:
    def __init__(self, dimension=None, max_concentration=500, spline_markers=1000):
        self.dimension = dimension
        self.max_concentration = max_concentration
        self.spline_markers = spline_markers

    def hypergeometric_ratio(self, concentration):
        """Largest covariance eigenvalue as a function of kappa (reference :258-262)."""
        from scipy.special import hyp1f1
        return hyp1f1(2, self.dimension + 1, concentration) / (
            self.dimension * hyp1f1(1, self.dimension, concentration))

    @cached_property
    def spline(self):
        """Quadratic interp1d of the INVERSE ratio on a log grid (reference :238-256)."""
        from scipy.interpolate import interp1d
        assert self.dimension is not None, (
            'You need to specify dimension. This can be done at object '
            'instantiation or it can be inferred when using the fit function.')
        x = np.logspace(-3, np.log10(self.max_concentration), self.spline_markers)
        y = self.hypergeometric_ratio(x)
        return interp1d(y, x, kind='quadratic', assume_sorted=True, bounds_error=False,
                        fill_value=(0, self.max_concentration))

    def hypergeometric_ratio_inverse(self, eigenvalues):
        """kappa(eigenvalue), host evaluation (reference :264-273)."""
        return self.spline(eigenvalues)

    def device_spline(self, device=None):
        """Knots / coefficients of the same spline for the kernel's de Boor evaluation."""
        t = _lib.torch()
        bs = self.spline._spline  # scipy.interpolate.BSpline, k = 2
        assert bs.k == 2, bs.k
        return dict(t=_lib.to_device(np.ascontiguousarray(bs.t), t.float64, device),
                    c=_lib.to_device(np.ascontiguousarray(bs.c).ravel(), t.float64, device),
                    ev_min=float(self.spline.x[0]), ev_max=float(self.spline.x[-1]),
                    max_concentration=float(self.max_concentration))

    def fit(self, y, saliency=None):
        """y (..., N, D) -> ComplexWatson (reference :274-298): one M-step of the
        mixture kernel with a single class and the saliency as affiliation."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert y.dtype in (t.complex64, t.complex128), y.dtype
        assert y.shape[-1] > 1
        *indep, N, D = y.shape
        if self.dimension is None:
            self.dimension = D
        else:
            assert self.dimension == D, (
                'You initialized the trainer with a different dimension than '
                'you are using to fit a model. Use a new trainer, when you '
                'change the dimension.')
        yb = y.reshape(-1, N, D).contiguous()
        B = yb.shape[0]
        if saliency is None:
            g0 = t.ones((B, 1, N), dtype=t.float64, device=yb.device)
        else:
            g0 = _lib.to_device(saliency, t.float64).to(yb.device).expand(*indep, N)
            g0 = g0.reshape(B, 1, N).contiguous()
        r = engine.cwmm_fit(yb, 1, self.device_spline(yb.device), gamma0=g0, iterations=1)
        return ComplexWatson(
            mode=as_result(r['mode'].reshape(*indep, D), like_torch),
            concentration=as_result(r['concentration'].reshape(*indep), like_torch))
