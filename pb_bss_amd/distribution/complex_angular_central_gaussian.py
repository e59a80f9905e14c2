"""Complex angular central Gaussian (cACG) on the device.

Mirrors pb_bss/distribution/complex_angular_central_gaussian.py: same class
and function names, argument meaning, shapes and error behaviour; the bodies
call the HIP library instead of NumPy/LAPACK.
"""
from dataclasses import dataclass

import numpy as np

from .. import _lib, engine
from .utils import _ProbabilisticModel, as_result
from .utils import force_hermitian  # noqa: F401  (names the reference module exposes)
from ..utils import is_broadcast_compatible  # noqa: F401  (names the reference module exposes)

__all__ = [
    'ComplexAngularCentralGaussian',
    'ComplexAngularCentralGaussianTrainer',
    'normalize_observation',
    'sample_complex_angular_central_gaussian',
]


def _complex_device(y):
    t = _lib.torch()
    y = _lib.to_device(y)
    if y.dtype not in (t.complex64, t.complex128):
        raise AssertionError(y.dtype)  # reference: assert np.iscomplexobj(y)
    return y


def normalize_observation(observation):
    """(..., N, D) -> unit norm over D -> contiguous (..., D, N).

    Reference: complex_angular_central_gaussian.py:34-55 (attention: swaps the
    D and N axes); zero vectors stay zero (utils.py:223-256, eps_style='where').
    """
    like_torch = _lib.is_torch(observation)
    y = _complex_device(observation)
    *indep, N, D = y.shape
    out = engine.normalize_observation(y.reshape(-1, N, D))
    return as_result(out.reshape(*indep, D, N), like_torch)


def sample_complex_angular_central_gaussian(size, covariance):
    """Draw `size` unit vectors from a cACG: a circularly-symmetric complex Gaussian
    sample with the given (D, D) covariance, projected to the unit sphere.  Host-side
    test-data utility (global NumPy RNG, real parts drawn before imaginary parts, so a
    seeded run reproduces the reference's draws).  Reference:
    complex_angular_central_gaussian.py:58-65, complex_circular_symmetric_gaussian.py:47-69."""
    covariance = np.asarray(_lib.to_host(covariance) if _lib.is_torch(covariance) else covariance)
    if covariance.ndim > 2:
        raise NotImplementedError(
            "Not quite clear how the correct broadcasting would look like.")
    D = covariance.shape[-1]
    re = np.random.normal(size=(*size, D))
    im = np.random.normal(size=(*size, D))
    x = (re + 1j * im) / np.sqrt(2)
    x = x @ np.linalg.cholesky(covariance).T
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def _broadcast_params(eigvec, eigval, indep, K):
    """Expand (..., K, D, D)/(..., K, D) parameters to the flattened batch."""
    t = _lib.torch()
    D = eigvec.shape[-1]
    vec = _lib.to_device(eigvec, t.complex128)
    val = _lib.to_device(eigval, t.float64)
    vec = vec.expand(*indep, K, D, D).reshape(-1, K, D, D).contiguous()
    val = val.expand(*indep, K, D).reshape(-1, K, D).contiguous()
    return vec, val


@dataclass
class ComplexAngularCentralGaussian(_ProbabilisticModel):
    """Eigen-parameterised cACG (reference :68-79): eigenvectors (..., D, D) in
    columns and (floored) eigenvalues (..., D)."""
    covariance_eigenvectors: np.ndarray = None  # (..., D, D)
    covariance_eigenvalues: np.ndarray = None  # (..., D)

    @classmethod
    def from_covariance(cls, covariance, eigenvalue_floor=0.,
                        covariance_norm='eigenvalue'):
        """Reference :82-132.  Hermitian eigendecomposition on the device
        (batched Jacobi instead of LAPACK zheevd), then the eigenvalue
        normalisation / floor of the reference."""
        like_torch = _lib.is_torch(covariance)
        t = _lib.torch()
        cov = _lib.to_device(covariance, t.complex128)
        *indep, D, D2 = cov.shape
        assert D == D2, cov.shape
        if covariance_norm == 'trace':
            tr = t.einsum('...dd', cov).real[..., None, None]
            cov = cov / t.clamp(tr, min=np.finfo(np.float64).tiny)
        else:
            assert covariance_norm in ['eigenvalue', False], covariance_norm
        val, vec, st = engine.heev(cov.reshape(-1, D, D).contiguous())
        if int(st.max().item()) & _lib.ST_EIG_NOCONV:
            if eigenvalue_floor == 0:
                raise RuntimeError(
                    'When you set the eigenvalue_floor to zero it can happen '
                    'that the eigenvalues get zero and the reciprocal '
                    f'eigenvalue that is used in {cls.__name__}._log_pdf gets '
                    'infinity.')
            raise np.linalg.LinAlgError('Eigenvalues did not converge')
        val = val.reshape(*indep, D)
        vec = vec.reshape(*indep, D, D)
        top = val.amax(dim=-1, keepdim=True)
        if covariance_norm == 'eigenvalue':
            val = val / t.clamp(top, min=np.finfo(np.float64).tiny)
            val = t.clamp(val, min=eigenvalue_floor)
        else:
            val = t.maximum(val, top * eigenvalue_floor)
        assert bool(t.isfinite(val).all()), val
        return cls(covariance_eigenvectors=as_result(vec, like_torch),
                   covariance_eigenvalues=as_result(val, like_torch))

    @property
    def covariance(self):
        """V diag(lambda) V^H (reference :141-148)."""
        v, lam = self.covariance_eigenvectors, self.covariance_eigenvalues
        if _lib.is_torch(v):
            t = _lib.torch()
            return t.einsum('...wx,...x,...zx->...wz', v, lam.to(v.dtype), v.conj())
        return np.einsum('...wx,...x,...zx->...wz', v, lam, v.conj())

    def sample(self, size):
        """Reference :134-138."""
        cov = self.covariance
        return sample_complex_angular_central_gaussian(size=size, covariance=cov)

    @property
    def log_determinant(self):
        lam = self.covariance_eigenvalues
        if _lib.is_torch(lam):
            return lam.log().sum(dim=-1)
        return np.sum(np.log(lam), axis=-1)

    def log_pdf(self, y):
        """y (..., N, D) -> log pdf (..., N) (reference :154-165)."""
        like_torch = _lib.is_torch(y)
        y = _complex_device(y)
        *indep, N, D = y.shape
        log_pdf, _ = self._device_log_pdf(y.reshape(-1, N, D), tuple(indep), N,
                                          layout=_lib.LAYOUT_TD)
        return as_result(log_pdf, like_torch)

    def _log_pdf(self, y):
        """y normalised (..., D, N) -> (log_pdf, quadratic_form), each
        (..., N) after broadcasting against the parameters (reference
        :167-203).  Mixture models call it with y[..., None, :, :]."""
        like_torch = _lib.is_torch(y)
        y = _complex_device(y)
        *indep, D, N = y.shape
        lp, q = self._device_log_pdf(y.reshape(-1, D, N), tuple(indep), N,
                                     layout=_lib.LAYOUT_DT)
        return as_result(lp, like_torch), as_result(q, like_torch)

    def _device_log_pdf(self, y_flat, indep, N, layout):
        t = _lib.torch()
        vec = _lib.to_device(self.covariance_eigenvectors, t.complex128)
        val = _lib.to_device(self.covariance_eigenvalues, t.float64)
        D = vec.shape[-1]
        # y independent axes (..., [1]) broadcast against parameter axes (..., K)
        p_indep = tuple(vec.shape[:-2])
        full = tuple(np.broadcast_shapes(indep, p_indep)) if (indep or p_indep) else ()
        # layout the problem as (B, K): K = trailing parameter axis that the
        # observation does not have (size 1 or missing), B = everything else
        if len(full) >= 1 and (len(indep) == 0 or indep[-1] == 1) and len(p_indep) >= 1:
            K = full[-1]
            lead = full[:-1]
            y_lead = indep[:-1] if indep else ()
        else:
            K = 1
            lead = full
            y_lead = indep
        B = int(np.prod(lead)) if lead else 1
        if layout == _lib.LAYOUT_TD:
            yb = y_flat.reshape(*y_lead, N, D).expand(*lead, N, D).reshape(B, N, D).contiguous()
        else:
            yb = y_flat.reshape(*y_lead, D, N).expand(*lead, D, N).reshape(B, D, N).contiguous()
        if K == 1 and full == lead:
            vb = vec.expand(*lead, D, D).reshape(B, 1, D, D).contiguous()
            lb = val.expand(*lead, D).reshape(B, 1, D).contiguous()
        else:
            vb = vec.expand(*lead, K, D, D).reshape(B, K, D, D).contiguous()
            lb = val.expand(*lead, K, D).reshape(B, K, D).contiguous()
        w = t.ones((B, K), dtype=t.float64, device=yb.device)
        _, q, lp = engine.em_predict(yb, vb, lb, w, layout=layout, want_q=True,
                                     want_log_pdf=True, want_affiliation=False)
        if K == 1 and full == lead:
            return lp.reshape(*full, N), q.reshape(*full, N)
        return lp.reshape(*lead, K, N), q.reshape(*lead, K, N)


class ComplexAngularCentralGaussianTrainer:
    def fit(self, y, saliency=None, hermitize=True, covariance_norm='eigenvalue',
            eigenvalue_floor=1e-10, iterations=10):
        """y (..., N, D) (reference :207-251): fixed-point iteration of the
        cACG maximum-likelihood covariance."""
        like_torch = _lib.is_torch(y)
        y = _complex_device(y)
        *indep, N, D = y.shape
        assert D > 1, y.shape
        if saliency is not None:
            raise NotImplementedError  # as the reference (:241-244)
        assert iterations > 0, iterations
        t = _lib.torch()
        yn = engine.normalize_observation(y.reshape(-1, N, D).to(t.complex128))  # (B, D, N), float64
        B = yn.shape[0]
        q = t.ones((B, 1, N), dtype=t.float64, device=yn.device)
        ones = t.ones((B, 1, N), dtype=t.float64, device=yn.device)
        for _ in range(iterations):
            vec, val, _, _ = engine.cacg_m_step(
                yn, ones, q, layout=_lib.LAYOUT_DT, covariance_norm=covariance_norm,
                eigenvalue_floor=eigenvalue_floor)
            w = t.ones((B, 1), dtype=t.float64, device=yn.device)
            _, q, _ = engine.em_predict(yn, vec, val, w, layout=_lib.LAYOUT_DT,
                                        want_q=True, want_affiliation=False)
        return ComplexAngularCentralGaussian(
            covariance_eigenvectors=as_result(vec.reshape(*indep, D, D), like_torch),
            covariance_eigenvalues=as_result(val.reshape(*indep, D), like_torch))

    def _fit(self, y, saliency, quadratic_form, hermitize=True,
             covariance_norm='eigenvalue', eigenvalue_floor=1e-10):
        """One M-step (reference :253-342).  y normalised (..., D, N) -- mixture
        models pass (..., 1, D, N); saliency (..., K, N) or None;
        quadratic_form (..., K, N)."""
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _complex_device(y)
        q = _lib.to_device(quadratic_form, t.float64)
        D, N = y.shape[-2:]
        *q_indep, Nq = q.shape
        assert Nq == N, (y.shape, q.shape)
        if saliency is None:
            sal = t.ones_like(q)
        else:
            sal = _lib.to_device(saliency, t.float64)
            assert y.ndim == sal.ndim + 1, (y.shape, sal.ndim)
            sal = sal.expand(*q_indep, N)
        # y independent axes broadcast against the (..., K) axes of q
        # (reference: is_broadcast_compatible(y.shape[:-2], q.shape[:-1]), :293-295)
        y_indep = tuple(y.shape[:-2])
        assert len(y_indep) <= len(q_indep), (y.shape, q.shape)
        y_indep = (1,) * (len(q_indep) - len(y_indep)) + y_indep
        y = y.reshape(*y_indep, D, N)
        if len(q_indep) >= 1 and y_indep[-1] == 1:
            lead, K = tuple(q_indep[:-1]), q_indep[-1]
            yb = y.reshape(*y_indep[:-1], D, N)
        else:
            lead, K = tuple(q_indep), 1
            yb = y
        B = int(np.prod(lead)) if lead else 1
        yb = yb.expand(*lead, D, N).reshape(B, D, N).contiguous()
        qb = q.reshape(B, K, N).contiguous()
        sb = sal.reshape(B, K, N).contiguous()
        vec, val, _, _ = engine.cacg_m_step(
            yb, sb, qb, layout=_lib.LAYOUT_DT, covariance_norm=covariance_norm,
            eigenvalue_floor=eigenvalue_floor)
        shape = (*lead, K) if (len(q_indep) >= 1 and y_indep[-1] == 1) else lead
        return ComplexAngularCentralGaussian(
            covariance_eigenvectors=as_result(vec.reshape(*shape, D, D), like_torch),
            covariance_eigenvalues=as_result(val.reshape(*shape, D), like_torch))
