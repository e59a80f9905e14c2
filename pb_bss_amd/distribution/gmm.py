"""Gaussian mixture model on real embeddings on the HIP embedding kernels.
Mirrors pb_bss/distribution/gmm.py:16-171: `GMM` (weight, gaussian; predict) and
`GMMTrainer` (fit / fit_predict) for covariance_type 'full' (the reference's default;
FP64 matrix-pipe kernels of csrc/gauss_full.hip, D <= 63) and 'spherical' (the covariance
model the joint GCACGMM uses, gcacgmm.py:141; the vMF mixture's E-step / M-step kernels on the
raw embedding): the whole EM loop is one C-ABI call (`pbbss_gmm_full_fit` / `pbbss_gmm_fit`).
covariance_type 'diagonal' and the `weight_constant_axis` sets beyond (-1,), (-2,), -2 run the
reference's loop step by step on the device (`_embed_stepwise.py`: log-pdf, softmax, weight and
single-Gaussian fit kernels, no host round trip inside the loop).  BinaryGMM wraps sklearn's
KMeans and is out of scope.
"""
from dataclasses import dataclass
from operator import xor

import numpy as np

from .. import _lib, engine
from .gaussian import DiagonalGaussian, Gaussian, SphericalGaussian
from .utils import _ProbabilisticModel, as_result, random_affiliation
from ..utils import labels_to_one_hot  # noqa: F401  (names the reference module exposes)
from .gaussian import GaussianTrainer  # noqa: F401  (names the reference module exposes)
from .mixture_model_utils import estimate_mixture_weight, log_pdf_to_affiliation  # noqa: F401  (names the reference module exposes)

__all__ = ['GMM', 'GMMTrainer']


_CLASS, _UNIFORM, _ONES = range(3)


def _weight_kind(weight_constant_axis, ndim):
    """How estimate_mixture_weight (mixture_model_utils.py:133-203) is driven:
    (-1,) / -1: per-class weights (K, 1); int -2: the constant 1/K (:180-183);
    tuple (-2,) -- the default of fit_predict --: the general path averages over the class
    axis and L1-normalises along it, i.e. a (1, N) array of ones (:192-201).  Weights that
    are constant over the classes cancel in the posterior, so both run as the uniform mode."""
    axis = weight_constant_axis
    if isinstance(axis, list):
        axis = tuple(axis)
    if isinstance(axis, int):
        if axis % ndim - ndim == -2:
            return _UNIFORM
        axis = (axis,)
    norm = tuple(a % ndim - ndim for a in axis)
    if norm == (-1,):
        return _CLASS
    if norm == (-2,):
        return _ONES
    return None  # any other axis set: the step-wise device loop (_embed_stepwise.py)


def _check_covariance_type(covariance_type):
    if covariance_type not in ('spherical', 'diagonal', 'full'):
        raise ValueError(f"Unknown covariance type '{covariance_type}'.")  # gaussian.py:184


_CLS = {'full': Gaussian, 'diagonal': DiagonalGaussian, 'spherical': SphericalGaussian}


def _kind_of(gaussian):
    for name, cls in _CLS.items():
        if type(gaussian) is cls:
            return name
    raise TypeError(type(gaussian))


@dataclass
class GMM(_ProbabilisticModel):
    weight: np.ndarray = None  # (..., K, 1)
    gaussian: object = None  # Gaussian (full covariance) or SphericalGaussian

    def predict(self, x):
        """x (..., N, D) real -> affiliations (..., K, N) (:21-25)."""
        like_torch = _lib.is_torch(x)
        t = _lib.torch()
        x = _lib.to_device(x)
        assert not x.is_complex(), x.dtype
        *indep, N, E = x.shape
        mean = _lib.to_device(self.gaussian.mean, t.float64).to(x.device)
        K = mean.shape[-2]
        cov = _lib.to_device(self.gaussian.covariance, t.float64).to(x.device)
        kind = _kind_of(self.gaussian)
        full = kind == 'full'
        w = _lib.to_device(self.weight, t.float64).to(x.device)
        general = kind == 'diagonal' or (w.shape[-1] != 1 and w.shape[-2] != 1) or (
            w.ndim > 2 and any(a != 1 for a in w.shape[:-2]) and w.shape[-1] != 1)
        if general:
            # diagonal covariances / weights that vary over classes AND frames: the general
            # log-pdf + softmax steps
            from . import _embed_stepwise as sw
            cs = {'full': (E, E), 'diagonal': (E,), 'spherical': ()}[kind]
            aff = sw.affiliation(
                kind, x.reshape(-1, N, E), mean.expand(*indep, K, E).reshape(-1, K, E).contiguous(),
                cov.expand(*indep, K, *cs).reshape(-1, K, *cs).contiguous(), w, tuple(indep), K, N)
            return as_result(aff.reshape(*indep, K, N), like_torch)
        if w.shape[-1] != 1:
            # (..., 1, N): constant over the classes (weight_constant_axis=(-2,)), cancels in
            # the posterior (mixture_model_utils.py:37-47) -- evaluated with uniform weights
            if w.shape[-2] != 1:
                raise NotImplementedError(f'frame-dependent class weights {tuple(w.shape)}')
            w = t.full((K, 1), 1.0 / K, dtype=t.float64, device=x.device)
        cshape = (E, E) if full else ()
        model = (mean.expand(*indep, K, E).reshape(-1, K, E).contiguous(),
                 cov.expand(*indep, K, *cshape).reshape(-1, K, *cshape).contiguous(),
                 w.expand(*indep, K, 1).reshape(-1, K).contiguous())
        if full:
            r = engine.gmm_full_fit(x.reshape(-1, N, E), K, model=model, iterations=0,
                                    final_predict=True)
            if int(r['status'].item()) != 0:
                raise ValueError(  # sklearn's _compute_precision_cholesky (gaussian.py:26)
                    'Fitting the mixture model failed because some components have ill-defined empirical '
                'covariance (not positive definite)')
        else:
            r = engine.gmm_fit(x.reshape(-1, N, E), K, model=model, iterations=0,
                               final_predict=True)
        return as_result(r['affiliation'].reshape(*indep, K, N), like_torch)


class GMMTrainer:
    def __init__(self, eps=1e-10):
        self.eps = eps
        self.log_likelihood_history = []

    def fit(self, y, initialization=None, num_classes=None, iterations=100, *, saliency=None,
            weight_constant_axis=(-1,), covariance_type='full', fixed_covariance=None):
        """y (..., N, D) real; initialization (..., K, N); saliency (..., N);
        fixed_covariance (..., K) (:33-95)."""
        assert xor(initialization is None, num_classes is None), (
            "Incompatible input combination. "
            "Exactly one of the two inputs has to be None: "
            f"{initialization is None} xor {num_classes is None}"
        )
        _check_covariance_type(covariance_type)
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        y = _lib.to_device(y)
        assert not y.is_complex(), y.dtype
        *indep, N, E = y.shape
        indep = tuple(indep)
        if initialization is None:
            # global NumPy RNG (:72-77)
            gamma0 = random_affiliation((*indep, num_classes, N), y.device)
        else:
            gamma0 = _lib.to_device(initialization, t.float64).to(y.device)
            num_classes = gamma0.shape[-2]
            gamma0 = gamma0.expand(*indep, num_classes, N)
        K = num_classes
        kind = _weight_kind(weight_constant_axis, len(indep) + 2)
        if kind is None or covariance_type == 'diagonal':
            if iterations <= 0:
                return None
            from . import _embed_stepwise as sw
            cs = {'full': (E, E), 'diagonal': (E,), 'spherical': ()}[covariance_type]
            fixed = None
            if fixed_covariance is not None:
                fixed = _lib.to_device(fixed_covariance, t.float64).to(y.device)
                assert tuple(fixed.shape) == (*indep, K, *cs), (
                    f'{tuple(fixed.shape)} != {(*indep, K, *cs)}')  # :161-163
            r = sw.fit(covariance_type, y, gamma0.contiguous(), iterations, saliency,
                       weight_constant_axis, fixed_scale=fixed)
            return GMM(
                weight=as_result(r['weight'], like_torch),
                gaussian=_CLS[covariance_type](
                    mean=as_result(r['mean'].reshape(*indep, K, E), like_torch),
                    covariance=as_result(r['scale'].reshape(*indep, K, *cs), like_torch)))
        mode = _lib.WEIGHT_PER_CLASS_MEAN if kind == _CLASS else _lib.WEIGHT_UNIFORM
        sal = None
        if saliency is not None:  # None: ones (:79-80), which the kernels assume anyway
            sal = _lib.to_device(saliency, t.float64).to(y.device).expand(*indep, N)
            sal = sal.reshape(-1, N).contiguous()
            if kind == _ONES and not bool((sal > 0).all().item()):
                raise NotImplementedError(
                    'weight_constant_axis=(-2,) with zero saliency entries (zero weights)')
        full = covariance_type == 'full'
        cshape = (E, E) if full else ()
        fixed = None
        if fixed_covariance is not None:
            fixed = _lib.to_device(fixed_covariance, t.float64).to(y.device)
            assert tuple(fixed.shape) == (*indep, K, *cshape), (
                f'{tuple(fixed.shape)} != {(*indep, K, *cshape)}')  # :161-163
            fixed = fixed.reshape(-1, K, *cshape).contiguous()
        if iterations <= 0:
            return None  # the reference's loop body never runs (:127-141)
        fit = engine.gmm_full_fit if full else engine.gmm_fit
        r = fit(y.reshape(-1, N, E), K, gamma0=gamma0.reshape(-1, K, N).contiguous(),
                iterations=iterations, saliency=sal, weight_mode=mode, fixed_covariance=fixed)
        if full and int(r['status'].item()) != 0:
            raise ValueError(  # sklearn's _compute_precision_cholesky (gaussian.py:26)
                'Fitting the mixture model failed because some components have ill-defined empirical '
                'covariance (not positive definite)')
        if kind == _UNIFORM:
            weight = t.full((K, 1), 1.0 / K, dtype=t.float64, device=y.device)
        elif kind == _ONES:
            weight = t.ones((*indep, 1, N), dtype=t.float64, device=y.device)
        else:
            weight = r['weight'].reshape(*indep, K, 1)
        cls = Gaussian if full else SphericalGaussian
        return GMM(
            weight=as_result(weight, like_torch),
            gaussian=cls(
                mean=as_result(r['mean'].reshape(*indep, K, E), like_torch),
                covariance=as_result(r['covariance'].reshape(*indep, K, *cshape), like_torch)))

    def fit_predict(self, y, initialization=None, num_classes=None, iterations=100, *,
                    saliency=None, weight_constant_axis=(-2,), covariance_type='full',
                    fixed_covariance=None):
        """Fit a model. Then just return the posterior affiliations (:97-119)."""
        model = self.fit(y=y, initialization=initialization, num_classes=num_classes,
                         iterations=iterations, saliency=saliency,
                         weight_constant_axis=weight_constant_axis,
                         covariance_type=covariance_type, fixed_covariance=fixed_covariance)
        return model.predict(y)
