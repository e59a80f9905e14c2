"""Shared host logic of the joint spatial + spectral mixtures (GCACGMM, VMFCACGMM):
argument handling around the single C-ABI call `pbbss_joint_fit` (csrc/capi.hip),
which enqueues, per EM iteration, the spectral log-pdf kernel, the joint cACG
E/M/eigen kernel, the class-weight reduction and the spectral M-step.
"""
import numpy as np

from .. import _lib, engine
from .complex_angular_central_gaussian import ComplexAngularCentralGaussian
from .utils import as_result, random_affiliation


import contextlib
import threading

_tls = threading.local()


@contextlib.contextmanager
def sharded_bins(enabled=True):
    """Inside this context `fit` treats the observation / embedding it is given as ONE RANK'S
    BLOCK of frequency bins: the spectral M-step sums and the bin-constant mixture weights are
    all-reduced over the library communicator (`pbbss_mix_opts.sharded`).  Used by
    `sharding.fit_predict_sharded_joint`."""
    old = getattr(_tls, 'sharded', False)
    _tls.sharded = bool(enabled)
    try:
        yield
    finally:
        _tls.sharded = old


def weight_mode(weight_constant_axis):
    """gcacgmm.py:158-162: axes refer to the (F, K, T) affiliation."""
    if isinstance(weight_constant_axis, int):
        weight_constant_axis = (weight_constant_axis,)
    if -2 in weight_constant_axis:  # the literal test of gcacgmm.py:288: (-3, -2, -1) -> 1/K
        return _lib.JOINT_WEIGHT_UNIFORM
    axes = tuple(sorted(a % 3 - 3 for a in weight_constant_axis))
    try:
        return {(-1,): _lib.JOINT_WEIGHT_FK, (-3, -1): _lib.JOINT_WEIGHT_K,
                (-3,): _lib.JOINT_WEIGHT_KT, (-3, -2, -1): _lib.JOINT_WEIGHT_CONST}[axes]
    except KeyError:
        raise ValueError(f'weight_constant_axis={weight_constant_axis!r}') from None


def prepare(observation, embedding):
    t = _lib.torch()
    obs = _lib.to_device(observation)
    emb = _lib.to_device(embedding).to(obs.device)
    assert obs.dtype in (t.complex64, t.complex128), obs.dtype
    assert not emb.is_complex(), emb.dtype
    assert obs.ndim == 3 and emb.ndim == 3 and obs.shape[:2] == emb.shape[:2], (obs.shape, emb.shape)
    return obs.contiguous(), emb.contiguous()


def initial_affiliation(initialization, num_classes, F, T, device):
    t = _lib.torch()
    if initialization is None:
        return random_affiliation((F, num_classes, T), device)  # global NumPy RNG, gcacgmm.py:186-190
    g = _lib.to_device(initialization, t.float64).to(device)
    assert g.shape[0] == F and g.shape[2] == T, (g.shape, F, T)
    return g.contiguous()


def fit(kind, observation, embedding, initialization, num_classes, iterations, saliency, *,
        covariance_norm, eigenvalue_floor, affiliation_eps, weight_constant_axis, spatial_weight,
        spectral_weight, inline_permutation_alignment, min_concentration=1e-10,
        max_concentration=500., fixed_scale=None):
    """-> (result dict of device tensors, like_torch)."""
    like_torch = _lib.is_torch(observation)
    t = _lib.torch()
    obs, emb = prepare(observation, embedding)
    assert obs.shape[-1] > 1
    F, T, _ = obs.shape
    gamma0 = initial_affiliation(initialization, num_classes, F, T, obs.device)
    K = gamma0.shape[1]
    assert iterations > 0, iterations
    sal = None
    if saliency is not None:
        sal = _lib.to_device(saliency, t.float64).to(obs.device).expand(F, T).contiguous()
    fixed = None
    if fixed_scale is not None:
        fixed = _lib.to_device(fixed_scale, t.float64).to(obs.device).contiguous()
        E = emb.shape[-1]
        want = {_lib.EMBED_GAUSS_FULL: (K, E, E), _lib.EMBED_GAUSS_DIAG: (K, E)}.get(kind, (K,))
        assert tuple(fixed.shape) == want, f'{tuple(fixed.shape)} != {want}'  # gcacgmm.py:306-308
    r = engine.joint_fit(
        obs, emb, K, kind, gamma0=gamma0, iterations=iterations, saliency=sal,
        weight_mode=weight_mode(weight_constant_axis),
        covariance_norm=_lib.COVNORM[covariance_norm], eigenvalue_floor=eigenvalue_floor,
        affiliation_eps=affiliation_eps, spatial_weight=spatial_weight,
        spectral_weight=spectral_weight, inline_pa=inline_permutation_alignment,
        min_concentration=min_concentration, max_concentration=max_concentration,
        fixed_scale=fixed, sharded=getattr(_tls, 'sharded', False))
    return r, like_torch


def predict(kind, model, spectral_mean, spectral_scale, observation, embedding):
    like_torch = _lib.is_torch(observation)
    t = _lib.torch()
    obs, emb = prepare(observation, embedding)
    F, T, D = obs.shape
    dev = obs.device
    vec = _lib.to_device(model.cacg.covariance_eigenvectors, t.complex128).to(dev).contiguous()
    val = _lib.to_device(model.cacg.covariance_eigenvalues, t.float64).to(dev).contiguous()
    K = vec.shape[1]
    mode = weight_mode(model.weight_constant_axis)
    w = _lib.to_device(np.asarray(model.weight) if not _lib.is_torch(model.weight) else model.weight,
                       t.float64).to(dev)
    w = w.reshape(engine.joint_weight_shape(mode, F, K, T)).contiguous()
    r = engine.joint_fit(
        obs, emb, K, kind,
        model=(vec, val, w, _lib.to_device(spectral_mean, t.float64).to(dev).contiguous(),
               _lib.to_device(spectral_scale, t.float64).to(dev).contiguous()),
        iterations=0, weight_mode=mode, spatial_weight=model.spatial_weight,
        spectral_weight=model.spectral_weight, final_predict=True)
    return as_result(r['affiliation'], like_torch)


def cacg_of(r, like_torch):
    return ComplexAngularCentralGaussian(
        covariance_eigenvectors=as_result(r['eigvec'], like_torch),
        covariance_eigenvalues=as_result(r['eigval'], like_torch))


def weight_of(r, mode, K, like_torch):
    """The reference stores a Python float for the uniform case (gcacgmm.py:289) and
    squeezed arrays otherwise (:295)."""
    if mode == _lib.JOINT_WEIGHT_UNIFORM:
        return 1 / K
    return as_result(r['weight'], like_torch)
