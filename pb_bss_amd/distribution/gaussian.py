"""Gaussians on real embeddings on the device.  Mirrors pb_bss/distribution/gaussian.py:
`SphericalGaussian` (:100-137; csrc/embed.hip -- the spectral half of GCACGMM, the default of
GCACGMMTrainer, gcacgmm.py:141), `Gaussian` with a full covariance (:17-56; csrc/gauss_full.hip:
weighted scatter on the FP64 matrix pipe, workgroup Cholesky, MFMA quadratic forms) and
`GaussianTrainer` (:140-193) for covariance_type 'spherical', 'diagonal' and 'full'.
`DiagonalGaussian` (:58-97) evaluates its log-pdf exactly as the reference writes it: the
(K, D) precisions are fed to einsum as ONE K x D matrix shared by all classes (:87-91), which is
what `GCACGMMTrainer(covariance_type='diagonal')` therefore optimises there -- and here.
"""
from dataclasses import dataclass

import numpy as np

from .. import _lib, engine
from .utils import _ProbabilisticModel, as_result
from ..utils import is_broadcast_compatible  # noqa: F401  (names the reference module exposes)

__all__ = ['Gaussian', 'DiagonalGaussian', 'SphericalGaussian', 'GaussianTrainer']


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
        if all(s == 1 for s in y_lead) and K <= 8:
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


@dataclass
class DiagonalGaussian(_ProbabilisticModel):
    mean: np.ndarray = None        # (K, D)
    covariance: np.ndarray = None  # (K, D)

    @property
    def precision_cholesky(self):
        c = self.covariance
        return c.rsqrt() if _lib.is_torch(c) else 1.0 / np.sqrt(c)

    @property
    def log_det_precision_cholesky(self):
        pc = self.precision_cholesky
        return pc.log().sum(-1) if _lib.is_torch(pc) else np.sum(np.log(pc), axis=-1)

    def log_pdf(self, y):
        """y (1, N, D) [or (N, D)] -> (K, N), as written in the reference (:76-97): the whitening
        uses the (K, D) precision array as one matrix, white[k, n, j] = sum_d pc[j, d] (y[n, d] -
        mean[k, d]) -- which, like there, needs a 2-D (K, D) model."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        N, E = y.shape[-2:]
        assert all(s == 1 for s in y.shape[:-2]), y.shape
        mean = _lib.to_device(self.mean, t.float64).to(y.device)
        cov = _lib.to_device(self.covariance, t.float64).to(y.device)
        assert mean.ndim == 2 and tuple(cov.shape) == tuple(mean.shape), (mean.shape, cov.shape)
        K = mean.shape[0]
        out = engine.embed_log_pdf(y.reshape(1, N, E), _lib.EMBED_GAUSS_DIAG,
                                   mean.reshape(1, K, E).contiguous(),
                                   cov.reshape(1, K, E).contiguous())
        return as_result(out.reshape(K, N), like_torch)


@dataclass
class Gaussian(_ProbabilisticModel):
    mean: np.ndarray = None        # (..., D)
    covariance: np.ndarray = None  # (..., D, D)

    @property
    def precision_cholesky(self):
        """sklearn's 'full' precision Cholesky (gaussian.py:26-30): P = L^-T for cov = L L^T."""
        cov = self.covariance
        if _lib.is_torch(cov):
            t = _lib.torch()
            chol = t.linalg.cholesky(cov)
            eye = t.eye(cov.shape[-1], dtype=cov.dtype, device=cov.device).expand_as(cov)
            return t.linalg.solve_triangular(chol, eye, upper=False).transpose(-1, -2)
        chol = np.linalg.cholesky(cov)
        return np.swapaxes(np.linalg.inv(chol), -1, -2)

    @property
    def log_det_precision_cholesky(self):
        pc = self.precision_cholesky
        d = pc.diagonal(dim1=-2, dim2=-1) if _lib.is_torch(pc) else np.diagonal(pc, axis1=-2, axis2=-1)
        return d.log().sum(-1) if _lib.is_torch(pc) else np.sum(np.log(d), axis=-1)

    def log_pdf(self, y):
        """y (..., N, D) -> (..., N), evaluated as the reference writes it (:35-56)."""
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
        if all(s == 1 for s in y_lead) and K <= 64:
            out, st = engine.gauss_full_log_pdf(y.reshape(1, N, E),
                                                mean.reshape(1, K, E).contiguous(),
                                                cov.reshape(1, K, E, E).contiguous())
            shape = lead
        else:
            shape = np.broadcast_shapes(y_lead, lead)
            out, st = engine.gauss_full_log_pdf(
                y.expand(*shape, N, E).reshape(-1, N, E).contiguous(),
                mean.expand(*shape, E).reshape(-1, 1, E).contiguous(),
                cov.expand(*shape, E, E).reshape(-1, 1, E, E).contiguous())
        if int(st.item()) != 0:
            raise ValueError(  # sklearn's _compute_precision_cholesky (gaussian.py:26)
                'Fitting the mixture model failed because some components have ill-defined empirical '
                'covariance (not positive definite)')
        return as_result(out.reshape(*shape, N), like_torch)


class GaussianTrainer:
    def fit(self, y, saliency=None, covariance_type='full'):
        """y (..., N, D) real, saliency (..., N) (:140-150)."""
        return self._fit(y, saliency=saliency, covariance_type=covariance_type)

    def _fit(self, y, saliency, covariance_type):
        """(:152-193) for covariance_type 'spherical', 'diagonal' and 'full'."""
        if covariance_type not in ('spherical', 'diagonal', 'full'):
            raise ValueError(f"Unknown covariance type '{covariance_type}'.")
        if covariance_type == 'full':
            return self._fit_full(y, saliency)
        kind = (_lib.EMBED_GAUSS_DIAG if covariance_type == 'diagonal'
                else _lib.EMBED_GAUSS_SPHERICAL)
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
        if all(s == 1 for s in y.shape[:-2]) and K <= 8:
            mean, cov = engine.embed_fit(y.reshape(1, N, E), kind,
                                         sal.expand(*lead, N).reshape(1, K, N).contiguous())
        else:
            mean, cov = engine.embed_fit(
                y.expand(*lead, N, E).reshape(-1, N, E).contiguous(), kind,
                sal.expand(*lead, N).reshape(-1, 1, N).contiguous())
        if covariance_type == 'diagonal':
            return DiagonalGaussian(mean=as_result(mean.reshape(*lead, E), like_torch),
                                    covariance=as_result(cov.reshape(*lead, E), like_torch))
        return SphericalGaussian(mean=as_result(mean.reshape(*lead, E), like_torch),
                                 covariance=as_result(cov.reshape(lead), like_torch))

    def _fit_full(self, y, saliency):
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
        if all(s == 1 for s in y.shape[:-2]):
            mean, cov = engine.gauss_full_fit(y.reshape(1, N, E),
                                              sal.expand(*lead, N).reshape(1, K, N).contiguous())
        else:
            mean, cov = engine.gauss_full_fit(
                y.expand(*lead, N, E).reshape(-1, N, E).contiguous(),
                sal.expand(*lead, N).reshape(-1, 1, N).contiguous())
        return Gaussian(mean=as_result(mean.reshape(*lead, E), like_torch),
                        covariance=as_result(cov.reshape(*lead, E, E), like_torch))
