"""Spherical Gaussian on the HIP embedding kernels (csrc/embed.hip): the spectral
half of GCACGMM.  Mirrors pb_bss/distribution/gaussian.py:100-193 for
covariance_type='spherical' (the default of GCACGMMTrainer, gcacgmm.py:141).
'full' / 'diagonal' covariances are not on the device path of this round.
"""
from dataclasses import dataclass

import numpy as np

from .. import _lib, engine
from .utils import _ProbabilisticModel, as_result

__all__ = ['SphericalGaussian', 'GaussianTrainer']


@dataclass
class SphericalGaussian(_ProbabilisticModel):
    mean: np.ndarray = None        # (..., D)
    covariance: np.ndarray = None  # (...,)

    @property
    def precision_cholesky(self):
        """1 / sqrt(covariance) (sklearn's 'diag' precision Cholesky, :110-112)."""
        c = self.covariance
        return c.rsqrt() if _lib.is_torch(c) else 1.0 / np.sqrt(c)

    @property
    def log_det_precision_cholesky(self):
        D = self.mean.shape[-1]
        pc = self.precision_cholesky
        return D * (pc.log() if _lib.is_torch(pc) else np.log(pc))

    def log_pdf(self, y):
        """y (..., N, D) -> (..., N); class axes of the model against a singleton
        axis of y are evaluated as one mixture (:116-137)."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        N, E = y.shape[-2:]
        mean = _lib.to_device(self.mean, t.float64).to(y.device)
        cov = _lib.to_device(self.covariance, t.float64).to(y.device)
        lead = tuple(mean.shape[:-1])
        y_lead = tuple(y.shape[:-2])
        K = int(np.prod(lead)) if lead else 1
        if all(s == 1 for s in y_lead) and K <= 6:
            out = engine.embed_log_pdf(y.reshape(1, N, E), _lib.EMBED_GAUSS_SPHERICAL,
                                       mean.reshape(1, K, E).contiguous(),
                                       cov.reshape(1, K).contiguous())
            return as_result(out.reshape(*lead, N), like_torch)
        shape = np.broadcast_shapes(y_lead, lead)
        out = engine.embed_log_pdf(
            y.expand(*shape, N, E).reshape(-1, N, E).contiguous(), _lib.EMBED_GAUSS_SPHERICAL,
            mean.expand(*shape, E).reshape(-1, 1, E).contiguous(),
            cov.expand(*shape).reshape(-1, 1).contiguous())
        return as_result(out.reshape(*shape, N), like_torch)


class GaussianTrainer:
    def fit(self, y, saliency=None, covariance_type='full'):
        """y (..., N, D) real, saliency (..., N) (:140-150)."""
        return self._fit(y, saliency=saliency, covariance_type=covariance_type)

    def _fit(self, y, saliency, covariance_type):
        """(:152-193) for covariance_type='spherical'."""
        if covariance_type != 'spherical':
            if covariance_type in ('full', 'diagonal'):
                raise NotImplementedError(
                    f"covariance_type={covariance_type!r}: only 'spherical' runs on the device")
            raise ValueError(f"Unknown covariance type '{covariance_type}'.")
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        N, E = y.shape[-2:]
        if saliency is None:
            sal = t.ones(y.shape[:-1], dtype=t.float64, device=y.device)
        else:
            sal = _lib.to_device(saliency, t.float64).to(y.device)
        lead = np.broadcast_shapes(tuple(y.shape[:-2]), tuple(sal.shape[:-1]))
        K = int(np.prod(lead)) if lead else 1
        if all(s == 1 for s in y.shape[:-2]) and K <= 6:
            mean, cov = engine.embed_fit(y.reshape(1, N, E), _lib.EMBED_GAUSS_SPHERICAL,
                                         sal.expand(*lead, N).reshape(1, K, N).contiguous())
        else:
            mean, cov = engine.embed_fit(
                y.expand(*lead, N, E).reshape(-1, N, E).contiguous(), _lib.EMBED_GAUSS_SPHERICAL,
                sal.expand(*lead, N).reshape(-1, 1, N).contiguous())
        return SphericalGaussian(mean=as_result(mean.reshape(*lead, E), like_torch),
                                 covariance=as_result(cov.reshape(lead), like_torch))
