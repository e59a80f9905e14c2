"""von Mises-Fisher distribution and trainer on the HIP embedding kernels
(csrc/embed.hip).  Mirrors pb_bss/distribution/von_mises_fisher.py:28-144.
"""
from dataclasses import dataclass

import numpy as np

from .. import _lib, engine
from .utils import _ProbabilisticModel, as_result
from ..utils import is_broadcast_compatible  # noqa: F401  (names the reference module exposes)

__all__ = ['VonMisesFisher', 'VonMisesFisherTrainer']


def _flat_model(mean, scale, device):
    """(..., K, E), (..., K) -> (B, K, E), (B, K) float64 device tensors.  A model
    without independent axes ((K, E)) is one mixture."""
    t = _lib.torch()
    mean = _lib.to_device(mean, t.float64).to(device)
    scale = _lib.to_device(scale, t.float64).to(device)
    if mean.ndim == 1:
        mean, scale = mean[None], scale.reshape(1)
    K, E = mean.shape[-2:]
    return mean.reshape(-1, K, E).contiguous(), scale.reshape(-1, K).contiguous()


@dataclass
class VonMisesFisher(_ProbabilisticModel):
    mean: np.ndarray = None           # (..., D)
    concentration: np.ndarray = None  # (...,)

    def log_pdf(self, y):
        """y (..., N, D) -> (..., N) with the model axes broadcast as the reference does
        (:62-78): the leading axes of `mean` are class axes of ONE set of observations
        when y has a singleton there, e.g. y (1, N, D), mean (K, D) -> (K, N)."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        N, E = y.shape[-2:]
        mean = _lib.to_device(self.mean, t.float64)
        lead = tuple(mean.shape[:-1])
        y_lead = tuple(y.shape[:-2])
        if all(s == 1 for s in y_lead):
            # every model is evaluated on the same N points: one mixture with prod(lead) classes
            K = int(np.prod(lead)) if lead else 1
            if K <= 8:
                out = engine.embed_log_pdf(
                    y.reshape(1, N, E), _lib.EMBED_VMF,
                    mean.reshape(1, K, E).to(y.device).contiguous(),
                    _lib.to_device(self.concentration, t.float64).reshape(1, K).to(y.device).contiguous())
                return as_result(out.reshape(*lead, N), like_torch)
        # general case: independent axes of y pair with those of the model
        shape = np.broadcast_shapes(y_lead, lead)
        yb = y.expand(*shape, N, E).reshape(-1, N, E).contiguous()
        mb = mean.to(y.device).expand(*shape, E).reshape(-1, 1, E).contiguous()
        cb = _lib.to_device(self.concentration, t.float64).to(y.device).expand(*shape)
        out = engine.embed_log_pdf(yb, _lib.EMBED_VMF, mb, cb.reshape(-1, 1).contiguous())
        return as_result(out.reshape(*shape, N), like_torch)

    def log_norm(self):
        """(:33-44) by the device series of csrc/embed.hip, read back through
        log_pdf(mean) = kappa * |mean| - log_norm."""
        like_torch = _lib.is_torch(self.mean)
        t = _lib.torch()
        mean = _lib.to_device(self.mean, t.float64)
        E = mean.shape[-1]
        lead = tuple(mean.shape[:-1])
        conc = _lib.to_device(self.concentration, t.float64).to(mean.device)
        m2 = mean.reshape(-1, 1, E).contiguous()
        lp = engine.embed_log_pdf(m2, _lib.EMBED_VMF, m2, conc.reshape(-1, 1).contiguous())
        out = conc.reshape(-1) * t.linalg.vector_norm(m2[:, 0], dim=-1) - lp.reshape(-1)
        return as_result(out.reshape(lead), like_torch)

    def pdf(self, y):
        lp = self.log_pdf(y)
        return lp.exp() if _lib.is_torch(lp) else np.exp(lp)


class VonMisesFisherTrainer:
    def fit(self, y, saliency=None, min_concentration=1e-10, max_concentration=500):
        """y (..., N, D) real, saliency (..., N) or None (:85-117)."""
        y_t = _lib.to_device(y)
        assert not y_t.is_complex(), y_t.dtype
        return self._fit(y, saliency=saliency, min_concentration=min_concentration,
                         max_concentration=max_concentration, _normalize=True)

    def _fit(self, y, saliency, min_concentration, max_concentration, _normalize=False):
        """(:119-144).  saliency may carry extra leading class axes ((K, N) against
        y (1, N, D)) exactly as the mixture trainers call it."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        N, E = y.shape[-2:]
        if saliency is None:
            sal = t.ones(y.shape[:-1], dtype=t.float64, device=y.device)
        else:
            sal = _lib.to_device(saliency, t.float64).to(y.device)
        lead = np.broadcast_shapes(tuple(y.shape[:-2]), tuple(sal.shape[:-1]))
        if all(s == 1 for s in y.shape[:-2]) and int(np.prod(lead)) <= 8:
            K = int(np.prod(lead)) if lead else 1
            mean, conc = engine.embed_fit(
                y.reshape(1, N, E), _lib.EMBED_VMF,
                sal.expand(*lead, N).reshape(1, K, N).contiguous(), normalize=_normalize,
                min_concentration=min_concentration, max_concentration=max_concentration)
        else:
            yb = y.expand(*lead, N, E).reshape(-1, N, E).contiguous()
            mean, conc = engine.embed_fit(
                yb, _lib.EMBED_VMF, sal.expand(*lead, N).reshape(-1, 1, N).contiguous(),
                normalize=_normalize, min_concentration=min_concentration,
                max_concentration=max_concentration)
        return VonMisesFisher(mean=as_result(mean.reshape(*lead, E), like_torch),
                              concentration=as_result(conc.reshape(lead), like_torch))
