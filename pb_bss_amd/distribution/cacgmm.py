"""cACGMM model and EM trainer backed by the persistent HIP EM kernel.

Mirrors pb_bss/distribution/cacgmm.py: `CACGMM` (predict / log_likelihood)
and `CACGMMTrainer` (fit / fit_predict) with the reference's argument names,
shapes (y is (..., N, D); affiliations (..., K, N)) and assertions.

Three execution paths:
  * fused   -- the whole EM loop in ONE kernel launch (pbbss_cacgmm_fit).  Used
               whenever every independent problem is self-contained:
               weight_constant_axis in {(-1,), -1, -2}, no inline aligner.
  * shared  -- weight_constant_axis (-3,) / (-3, -1) (weights averaged over the
               frequency bins): still one launch, a cooperative one whose workgroups
               exchange their affiliations once per iteration (pbbss_cacgmm_fit_shared).
  * stepwise -- E-step / hook / M-step per iteration (pbbss_cacgmm_predict,
               pbbss_estimate_mixture_weight, pbbss_cacg_m_step) for everything else that
               couples frequencies: inline_permutation_aligner (cacgmm.py:260-267), other
               axis sets, more bins than fit the device at once.
Arithmetic is float64 on the device for complex64 and complex128 input alike
(SURVEY.md section 7); outputs are float64 / complex128.
"""
from dataclasses import dataclass, field
from operator import xor

import numpy as np

from .. import _lib, engine
from .complex_angular_central_gaussian import (
    ComplexAngularCentralGaussian,
    ComplexAngularCentralGaussianTrainer,
    _complex_device,
    normalize_observation,
)
from .mixture_model_utils import (  # noqa: F401  (log_pdf_to_affiliation: the reference re-exports it)
    _host_estimate_mixture_weight,
    apply_inline_permutation_alignment,
    estimate_mixture_weight,
    log_pdf_to_affiliation,
)
from .utils import (_ProbabilisticModel, as_result, random_affiliation, reference_arithmetic,
                    reference_single, to_single)

__all__ = ['CACGMM', 'CACGMMTrainer', 'normalize_observation', 'sample_cacgmm']


def sample_cacgmm(size, weight, covariance, return_label=False):
    """`size` draws from a cACG mixture with class probabilities `weight` (K,) and class
    covariances (K, D, D); host-side test-data utility on the global NumPy RNG (labels
    first, then the classes in order).  Reference: cacgmm.py:27-55."""
    from .complex_angular_central_gaussian import sample_complex_angular_central_gaussian
    weight = np.asarray(weight)
    covariance = np.asarray(covariance)
    assert weight.ndim == 1, weight
    assert isinstance(size, int), size
    assert covariance.ndim == 3, covariance.shape
    num_classes, = weight.shape
    D = covariance.shape[-1]
    assert covariance.shape == (num_classes, D, D), (covariance.shape, num_classes, D)
    labels = np.random.choice(range(num_classes), size=size, p=weight)
    x = np.zeros((size, D), dtype=np.complex128)
    for k in range(num_classes):
        # the reference samples from the eigen-normalised model (from_covariance, :49-52):
        # the direction distribution only depends on the covariance up to scale, but the
        # Cholesky factor (and with it the exact draws) does not -- restate that scaling.
        lam, vec = np.linalg.eigh(covariance[k])
        lam = lam / np.maximum(np.amax(lam, axis=-1, keepdims=True), np.finfo(lam.dtype).tiny)
        cov_k = (vec * lam[..., None, :]) @ vec.conj().T
        x[labels == k, :] = sample_complex_angular_central_gaussian(
            size=(int(np.sum(labels == k)),), covariance=cov_k)
    return (x, labels) if return_label else x


def _weight_for_predict(weight, indep, K, N, device):
    """Expand a reference-shaped weight ((..., K, 1), (K, 1), (..., 1, K, N), ...)
    to (B, K) or (B, K, N) float64 on the device."""
    t = _lib.torch()
    w = _lib.to_device(weight, t.float64).to(device)
    if w.shape[-1] == 1:
        return w.expand(*indep, K, 1).reshape(-1, K).contiguous()
    return w.expand(*indep, K, N).reshape(-1, K, N).contiguous()


def _activity(mask, indep, K, N, device):
    if mask is None:
        return None
    t = _lib.torch()
    is_bool = (mask.dtype == t.bool) if _lib.is_torch(mask) else (mask.dtype == bool)
    assert is_bool, mask.dtype  # reference: mixture_model_utils.py:40
    m = _lib.to_device(mask).to(device).to(t.uint8)
    assert tuple(m.shape[-2:]) == (K, N), (m.shape, K, N)
    return m.expand(*indep, K, N).reshape(-1, K, N).contiguous()


def _stepwise_graph_enabled():
    # opt-in: measured on an MI355X (profiles/r05_i_stepwise_graph.txt) one graph launch per
    # iteration is NOT faster than the dozen eager launches it replaces -- the loop is bound by the
    # device time of its kernels, not by launch overhead, and capture + instantiation cost ~0.7 ms
    import os
    return os.environ.get('PBBSS_STEPWISE_GRAPH', '0') == '1'


@dataclass
class CACGMM(_ProbabilisticModel):
    """weight (..., K, 1) [or (K, 1) / (..., 1, K, N)], cacg parameters with a
    class axis (..., K, D, D) / (..., K, D).  Reference: cacgmm.py:58-62."""
    weight: np.ndarray = None
    cacg: ComplexAngularCentralGaussian = field(
        default_factory=ComplexAngularCentralGaussian)

    def predict(self, y, return_quadratic_form=False, source_activity_mask=None):
        """y (..., N, D) -> affiliation (..., K, N); affiliation_eps = 0.
        Reference: cacgmm.py:64-71."""
        like_torch = _lib.is_torch(y)
        single = self._single_with(y)
        y = _complex_device(y)
        *indep, N, D = y.shape
        aff, q, _ = self._device_e_step(y.reshape(-1, N, D), tuple(indep), N,
                                        _lib.LAYOUT_TD, source_activity_mask,
                                        0.0, want_q=return_quadratic_form)
        if single:
            aff, q = to_single(aff), to_single(q)
        if return_quadratic_form:
            return as_result(aff, like_torch), as_result(q, like_torch)
        return as_result(aff, like_torch)

    def _single_with(self, y):
        """Result dtype 'reference': the posterior of single-precision observations under a
        single-precision cACG is float32 whatever the weight's dtype -- the reference
        multiplies the weight in place (mixture_model_utils.py:37)."""
        return reference_single(y, self.cacg.covariance_eigenvectors,
                                self.cacg.covariance_eigenvalues)

    def _predict(self, y, source_activity_mask=None, affiliation_eps=0.):
        """y normalised (..., D, N).  Returns (affiliation, quadratic_form,
        log_pdf), each (..., K, N).  Reference: cacgmm.py:73-95."""
        like_torch = _lib.is_torch(y)
        single = self._single_with(y)
        y = _complex_device(y)
        *indep, D, N = y.shape
        aff, q, lp = self._device_e_step(y.reshape(-1, D, N), tuple(indep), N,
                                         _lib.LAYOUT_DT, source_activity_mask,
                                         affiliation_eps, want_q=True,
                                         want_log_pdf=True)
        if single:
            aff, q, lp = to_single(aff), to_single(q), to_single(lp)
        return (as_result(aff, like_torch), as_result(q, like_torch),
                as_result(lp, like_torch))

    def _device_e_step(self, y_flat, indep, N, layout, source_activity_mask,
                       affiliation_eps, want_q=False, want_log_pdf=False):
        t = _lib.torch()
        vec = _lib.to_device(self.cacg.covariance_eigenvectors, t.complex128)
        val = _lib.to_device(self.cacg.covariance_eigenvalues, t.float64)
        K, D = vec.shape[-3], vec.shape[-1]
        vb = vec.expand(*indep, K, D, D).reshape(-1, K, D, D).contiguous()
        lb = val.expand(*indep, K, D).reshape(-1, K, D).contiguous()
        w = _weight_for_predict(self.weight, indep, K, N, y_flat.device)
        act = _activity(source_activity_mask, indep, K, N, y_flat.device)
        aff, q, lp = engine.em_predict(
            y_flat, vb, lb, w, activity=act, layout=layout,
            affiliation_eps=affiliation_eps, want_q=want_q,
            want_log_pdf=want_log_pdf)
        shape = (*indep, K, N)
        return (aff.reshape(shape), None if q is None else q.reshape(shape),
                None if lp is None else lp.reshape(shape))

    def log_likelihood(self, y):
        """sum over time-frequency of logsumexp_k(log_pdf) -- the reference
        (cacgmm.py:97-138) does not apply the mixture weights here."""
        t = _lib.torch()
        y = _complex_device(y)
        *indep, N, D = y.shape
        _, _, lp = self._device_e_step(y.reshape(-1, N, D), tuple(indep), N,
                                       _lib.LAYOUT_TD, None, 0.0,
                                       want_log_pdf=True)
        return np.float64(t.logsumexp(lp, dim=-2).sum().item())


class CACGMMTrainer:
    def fit(
            self,
            y,
            initialization=None,
            num_classes=None,
            iterations=100,
            *,
            saliency=None,
            source_activity_mask=None,
            weight_constant_axis=(-1,),
            hermitize=True,
            covariance_norm='eigenvalue',
            affiliation_eps=1e-10,
            eigenvalue_floor=1e-10,
            inline_permutation_aligner=None,
            _with_affiliation=False,
            _weight_hook=None,
    ):
        """Same contract as the reference (cacgmm.py:142-280).

        y: (..., N, D) complex.  initialization: affiliations (..., K, N) in
        [0, 1] or a CACGMM; otherwise num_classes (random initialisation from
        the global NumPy RNG exactly as the reference, cacgmm.py:208-209).
        Returns a CACGMM (NumPy fields for NumPy input, torch for torch input).
        """
        assert xor(initialization is None, num_classes is None), (
            "Incompatible input combination. "
            "Exactly one of the two inputs has to be None: "
            f"{initialization is None} xor {num_classes is None}"
        )
        like_torch = _lib.is_torch(y)
        t = _lib.torch()
        # arithmetic 'reference': a complex64 observation with an array initialisation is what the
        # reference computes in single precision (cacgmm.py:226-227) -> packed-FP32 kernel
        packed32 = (reference_arithmetic() and initialization is not None
                    and not isinstance(initialization, CACGMM)
                    and str(y.dtype).rsplit('.', 1)[-1] == 'complex64')
        # result dtype 'reference': an array initialisation is cast to y.real.dtype
        # (cacgmm.py:226-227), a model keeps its own, random affiliations are float64
        # (:208); a saliency enters the M step as it is (:336-339)
        if isinstance(initialization, CACGMM):
            single = reference_single(y, initialization.cacg.covariance_eigenvectors,
                                      initialization.cacg.covariance_eigenvalues, saliency)
        else:
            single = initialization is not None and reference_single(y, saliency)
        y = _complex_device(y)
        assert y.shape[-1] > 1, y.shape
        assert iterations > 0, iterations
        *indep, N, D = y.shape
        indep = tuple(indep)
        dev = y.device

        model = None
        gamma0 = None
        if initialization is None:
            assert num_classes is not None, num_classes
            shape = (*indep, num_classes, N)
            # global NumPy RNG, as the reference (utils.random_affiliation)
            gamma0 = random_affiliation(shape, dev)
        elif isinstance(initialization, CACGMM):
            num_classes = initialization.cacg.covariance_eigenvectors.shape[-3]
            model = initialization
        elif isinstance(initialization, np.ndarray) or _lib.is_torch(initialization):
            num_classes = initialization.shape[-2]
            assert num_classes > 1, num_classes
            shape = (*indep, num_classes, N)
            assert initialization.ndim == len(shape), (initialization.shape, shape)
            assert tuple(initialization.shape[-2:]) == shape[-2:], (
                initialization.shape, shape)
            gamma0 = _lib.to_device(initialization, t.float64).to(dev).expand(shape)
        else:
            raise TypeError('No sufficient initialization.')
        K = num_classes

        if isinstance(weight_constant_axis, list):
            weight_constant_axis = tuple(weight_constant_axis)
        if source_activity_mask is not None:
            assert tuple(source_activity_mask.shape[-2:]) == (K, N), (
                source_activity_mask.shape, indep, K, N)
            if gamma0 is not None and not isinstance(initialization, CACGMM) \
                    and initialization is not None:
                assert tuple(source_activity_mask.shape) == tuple(initialization.shape), (
                    source_activity_mask.shape, initialization.shape)
        assert K < 20, f'num_classes: {K}, sure?'
        assert D < 35, f'Channels: {D}, sure?'
        # (D = 33, 34 -- admitted by the sanity assert above -- run on 36-wide padded tiles of the
        # generic-size kernels since round 6: csrc/generic.hip)

        ndim = len(indep) + 2
        mode = self._weight_mode(weight_constant_axis, ndim)
        # _weight_hook (sharding.shared_weight_allreduce): the mixture weights are estimated over
        # bins that live on other ranks too -> step-wise loop with the hook between E and M
        fused = (mode is not None and inline_permutation_aligner is None and _weight_hook is None)
        if fused and model is not None:
            # a resumed model must carry per-class weights the kernel can hold
            w = model.weight
            fused = (w is not None and w.shape[-1] == 1)

        act = _activity(source_activity_mask, indep, K, N, dev)
        sal = None
        if saliency is not None:
            sal = _lib.to_device(saliency, t.float64).to(dev).expand(*indep, N)
            sal = sal.reshape(-1, N).contiguous()

        if fused:
            if packed32 and D <= 8 and K <= 6:
                try:
                    return self._rounded(self._fit_fused(
                        y.reshape(-1, N, D), indep, K, gamma0, model, iterations, sal,
                        act, mode, covariance_norm, affiliation_eps, eigenvalue_floor,
                        hermitize, like_torch, final_predict=_with_affiliation,
                        precision='f32'), single, mode)
                except NotImplementedError:
                    pass  # e.g. an utterance too long for the LDS-resident kernel: float64 serves it
                except (engine.StatusAssertionError, engine.StatusLinAlgError) as e:
                    # (only the status-derived errors: an argument check that fails in
                    # _fit_fused must surface, not run the fit twice)
                    # a status failure of the reduced-precision kernel (ill-conditioned bins are
                    # likelier to go non-finite in float32): the float64 kernel is the accuracy
                    # superset -- let it try before the reference's error is raised
                    import warnings
                    warnings.warn(f'packed-FP32 fit failed ({type(e).__name__}: {e}); repeating it '
                                  'in float64', RuntimeWarning)
            return self._rounded(self._fit_fused(
                y.reshape(-1, N, D), indep, K, gamma0, model, iterations, sal,
                act, mode, covariance_norm, affiliation_eps, eigenvalue_floor,
                hermitize, like_torch, final_predict=_with_affiliation), single, mode)
        smode = self._shared_mode(weight_constant_axis, ndim)
        if smode is not None and inline_permutation_aligner is None and _weight_hook is None:
            out = self._fit_shared(
                y.reshape(-1, N, D), indep, K, gamma0, model, iterations, sal, act, smode,
                covariance_norm, affiliation_eps, eigenvalue_floor, like_torch,
                final_predict=_with_affiliation)
            if out is not None:
                return self._rounded(out, single, mode)
        return self._rounded(self._fit_stepwise(
            y.reshape(-1, N, D), indep, K, gamma0, model, iterations, saliency,
            sal, act, weight_constant_axis, covariance_norm, affiliation_eps,
            eigenvalue_floor, hermitize, inline_permutation_aligner, like_torch,
            weight_hook=_weight_hook), single, mode)

    @staticmethod
    def _rounded(out, single, mode):
        """Result dtype 'reference' and single-precision operands: round the float64 results
        to what the reference returns (float32 weights / eigenvalues / masks, complex64
        eigenvectors; the constant 1/K weight of weight_constant_axis=-2 stays float64,
        mixture_model_utils.py:180-183)."""
        if not single:
            return out
        model, aff = out if isinstance(out, tuple) else (out, None)
        uniform = mode == _lib.WEIGHT_UNIFORM
        model = CACGMM(
            weight=model.weight if uniform else to_single(model.weight),
            cacg=ComplexAngularCentralGaussian(
                covariance_eigenvectors=to_single(model.cacg.covariance_eigenvectors),
                covariance_eigenvalues=to_single(model.cacg.covariance_eigenvalues)))
        if aff is None:
            return model
        return model, to_single(aff)

    @staticmethod
    def _weight_mode(axis, ndim):
        """Map weight_constant_axis to the kernel's built-in modes, or None."""
        if isinstance(axis, int):
            if axis % ndim - ndim == -2:
                return _lib.WEIGHT_UNIFORM  # mixture_model_utils.py:180-183
            axis = (axis,)
        axes = {a % ndim - ndim for a in axis}
        if axes == {-1}:
            return _lib.WEIGHT_PER_CLASS_MEAN
        return None

    @staticmethod
    def _shared_mode(axis, ndim):
        """weight_constant_axis that averages the weights over the last independent axis (the
        frequency bins): (-3,) and (-3, -1) run in the cooperative kernel."""
        if isinstance(axis, int):
            axis = (axis,)
        axes = {a % ndim - ndim for a in axis}
        if ndim >= 3 and axes == {-3}:
            return _lib.WEIGHT_SHARED_KT
        if ndim >= 3 and axes == {-3, -1}:
            return _lib.WEIGHT_SHARED_K
        return None

    # ----------------------------------------------------- weights shared over the bins
    def _fit_shared(self, yb, indep, K, gamma0, model, iterations, sal, act, smode,
                    covariance_norm, affiliation_eps, eigenvalue_floor, like_torch,
                    final_predict=False):
        """weight_constant_axis (-3,) / (-3, -1): the whole loop in one cooperative launch
        (pbbss_cacgmm_fit_shared); None if the configuration is not served there."""
        t = _lib.torch()
        B, N, D = yb.shape
        group = indep[-1]
        outer = tuple(indep[:-1])
        G = B // group
        Nw = N if smode == _lib.WEIGHT_SHARED_KT else 1
        dev_model = None
        g0 = None
        if model is not None:
            w = _lib.to_device(model.weight, t.float64).to(yb.device)
            try:
                w = w.expand(*outer, 1, K, Nw)
            except RuntimeError:
                return None  # weights of another shape: the step-wise loop takes any broadcast
            vec = _lib.to_device(model.cacg.covariance_eigenvectors, t.complex128).to(yb.device)
            val = _lib.to_device(model.cacg.covariance_eigenvalues, t.float64).to(yb.device)
            dev_model = (
                vec.expand(*indep, K, D, D).reshape(B, K, D, D).contiguous(),
                val.expand(*indep, K, D).reshape(B, K, D).contiguous(),
                w.reshape((G, K, N) if Nw == N else (G, K)).contiguous())
        else:
            g0 = gamma0.reshape(B, K, N).contiguous()
        r = engine.em_fit_shared(
            yb, K, group, weight_mode=smode, gamma0=g0, model=dev_model, iterations=iterations,
            saliency=sal, activity=act, covariance_norm=covariance_norm,
            affiliation_eps=affiliation_eps, eigenvalue_floor=eigenvalue_floor,
            final_predict=final_predict)
        if r is None:
            return None
        out = CACGMM(
            weight=as_result(r['weight'].reshape(*outer, 1, K, Nw), like_torch),
            cacg=ComplexAngularCentralGaussian(
                covariance_eigenvectors=as_result(
                    r['eigvec'].reshape(*indep, K, D, D), like_torch),
                covariance_eigenvalues=as_result(
                    r['eigval'].reshape(*indep, K, D), like_torch)))
        if final_predict:
            return out, as_result(r['affiliation'].reshape(*indep, K, N), like_torch)
        return out

    # ------------------------------------------------------------------ fused
    def _fit_fused(self, yb, indep, K, gamma0, model, iterations, sal, act, mode,
                   covariance_norm, affiliation_eps, eigenvalue_floor, hermitize,
                   like_torch, final_predict=False, precision='f64'):
        t = _lib.torch()
        B, N, D = yb.shape
        dev_model = None
        g0 = None
        if model is not None:
            vec = _lib.to_device(model.cacg.covariance_eigenvectors, t.complex128)
            val = _lib.to_device(model.cacg.covariance_eigenvalues, t.float64)
            w = _lib.to_device(model.weight, t.float64)
            dev_model = (
                vec.expand(*indep, K, D, D).reshape(B, K, D, D).contiguous(),
                val.expand(*indep, K, D).reshape(B, K, D).contiguous(),
                w.expand(*indep, K, 1).reshape(B, K).contiguous())
        else:
            g0 = gamma0.reshape(B, K, N).contiguous()
        r = engine.em_fit(
            yb, K, gamma0=g0, model=dev_model, iterations=iterations,
            saliency=sal, activity=act, covariance_norm=covariance_norm,
            weight_mode=mode, affiliation_eps=affiliation_eps,
            eigenvalue_floor=eigenvalue_floor, hermitize=hermitize,
            layout=_lib.LAYOUT_TD, final_predict=final_predict, precision=precision)
        if mode == _lib.WEIGHT_UNIFORM:
            weight = t.full((K, 1), 1.0 / K, dtype=t.float64, device=yb.device)
        else:
            weight = r['weight'].reshape(*indep, K, 1)
        out = CACGMM(
            weight=as_result(weight, like_torch),
            cacg=ComplexAngularCentralGaussian(
                covariance_eigenvectors=as_result(
                    r['eigvec'].reshape(*indep, K, D, D), like_torch),
                covariance_eigenvalues=as_result(
                    r['eigval'].reshape(*indep, K, D), like_torch)))
        if final_predict:
            return out, as_result(r['affiliation'].reshape(*indep, K, N), like_torch)
        return out

    # --------------------------------------------------------------- stepwise
    @staticmethod
    def _device_weight(aff, sal, weight_constant_axis, indep):
        """estimate_mixture_weight (mixture_model_utils.py:133-203) on the device for the
        axis sets that occur in practice: the trailing `r` independent axes and / or the frame
        axis.  aff (*indep, K, N) device tensor, sal (*indep, N) or None.
        Returns the reference-shaped weight (keepdims) as a device tensor, or None if the axis
        set is not of that form (the caller then takes the host formula)."""
        nd = len(indep) + 2
        axes = ((weight_constant_axis,) if isinstance(weight_constant_axis, int)
                else tuple(weight_constant_axis))
        axes = sorted({a % nd for a in axes})
        if nd - 2 in axes:  # the class axis: only the scalar form -2 is defined (handled earlier)
            return None
        red_n = (nd - 1) in axes
        ind_axes = [a for a in axes if a < nd - 2]
        r = len(ind_axes)
        if ind_axes != list(range(nd - 2 - r, nd - 2)):
            return None  # not a trailing block of independent axes
        K, N = aff.shape[-2:]
        Bi = int(np.prod(indep[len(indep) - r:], dtype=np.int64)) if r else 1
        Bo = int(np.prod(indep[:len(indep) - r], dtype=np.int64)) if len(indep) > r else 1
        a4 = aff.reshape(Bo, Bi, K, N).contiguous()
        s3 = None if sal is None else sal.reshape(Bo, Bi, N).contiguous()
        w = engine.estimate_mixture_weight(a4, s3, reduce_inner=r > 0, reduce_n=red_n)
        if w is None:  # not served (saliency with K > 16): the caller takes the host formula
            return None
        shape = list(indep[:len(indep) - r]) + [1] * r + [K, 1 if red_n else N]
        return w.reshape(shape)

    @staticmethod
    def _replay_iterations(t, e_step, m_step, vec, val, weight, m_status, aligner_status, count):
        """Capture one E + M iteration that reads the model from static tensors and writes the new
        model back into them, replay it `count` times.  -> (vec, val, weight) or None when the
        capture is refused (the caller continues eagerly).  The status words of the captured
        launches are OR-ed into static accumulators inside the graph and appended to the lists."""
        sv, sl, sw = vec.clone(), val.clone(), weight.clone()
        n_m, n_a = len(m_status), len(aligner_status)
        # status accumulators of the replays, shaped like the words of the last eager iteration
        acc_m = t.zeros_like(m_status[-1])
        acc_a = t.zeros_like(aligner_status[-1]) if aligner_status else None
        graph = t.cuda.CUDAGraph()
        try:
            with t.cuda.graph(graph):
                aff, q = e_step(sv, sl, sw)
                nv, nl, nw = m_step(aff, q)
                assert len(m_status) == n_m + 1 and len(aligner_status) <= n_a + 1
                acc_m.bitwise_or_(m_status[-1])
                if len(aligner_status) > n_a:
                    acc_a.bitwise_or_(aligner_status[-1])
                sv.copy_(nv)
                sl.copy_(nl)
                sw.copy_(nw)
        except Exception as e:  # noqa: BLE001 -- a runtime that cannot capture: eager loop
            import warnings
            warnings.warn(f'step-wise EM loop: graph capture refused ({type(e).__name__}: {e}); '
                          'continuing launch by launch', RuntimeWarning)
            del m_status[n_m:], aligner_status[n_a:]
            return None
        # the entries appended while capturing belong to the graph's memory pool: drop them
        del m_status[n_m:], aligner_status[n_a:]
        for _ in range(count):
            graph.replay()
        m_status.append(acc_m)
        if acc_a is not None:
            aligner_status.append(acc_a)
        # keep the graph (and its private memory pool, which owns the tensors above) alive until
        # the caller has read the results
        sv._pbbss_graph = graph
        return sv, sl, sw

    def _fit_stepwise(self, yb, indep, K, gamma0, model, iterations, saliency,
                      sal, act, weight_constant_axis, covariance_norm,
                      affiliation_eps, eigenvalue_floor, hermitize, aligner,
                      like_torch, weight_hook=None, _retry_team=True):
        """The reference loop (cacgmm.py:252-278) for the options that couple frequency bins
        (weight_constant_axis with independent axes, inline_permutation_aligner), one E-step
        and one M-step launch per iteration with the cross-bin reduction
        (`pbbss_estimate_mixture_weight`) and, optionally, the device permutation aligner in
        between.  Everything stays on the device: no host round trip inside the loop.
        (`hermitize` is accepted for signature compatibility: the device M-step accumulates the
        Hermitian-packed covariance, which is Hermitian by construction.)"""
        t = _lib.torch()
        B, N, D = yb.shape
        dev = yb.device
        if yb.dtype == t.complex64:
            # as in the fused fit: the kernels keep the raw complex64 frames (exact) and apply
            # 1 / |y|^2 in float64 -- half the bytes of a widened, normalised copy per launch
            yn, y_layout = yb.contiguous(), _lib.LAYOUT_TD  # (B, N, D)
        else:
            # normalise in float64 (the reference keeps the input precision)
            yn, y_layout = engine.normalize_observation(yb), _lib.LAYOUT_DT  # (B, D, N)
        shape = (*indep, K, N)
        sal_dev = None
        if saliency is not None:
            sal_dev = _lib.to_device(saliency, t.float64).to(dev).expand(*indep, N).contiguous()
        vec = val = weight = None
        if model is None:
            aff = gamma0.reshape(shape).to(t.float64).contiguous()
            q = t.ones(shape, dtype=t.float64, device=dev)
        else:
            vec = _lib.to_device(model.cacg.covariance_eigenvectors, t.complex128).to(dev)
            val = _lib.to_device(model.cacg.covariance_eigenvalues, t.float64).to(dev)
            weight = _lib.to_device(model.weight, t.float64).to(dev)
        device_aligner = aligner is not None and type(aligner).__module__.startswith('pb_bss_amd')
        aligner_status = []  # device status words of the aligner, read once after the loop
        m_status = []
        host_excursion = [not device_aligner and aligner is not None]  # any step that leaves the device

        def e_step(vec, val, weight):
            w = _weight_for_predict(weight, indep, K, N, dev)
            aff, q, _ = engine.em_predict(
                yn, vec.expand(*indep, K, D, D).reshape(B, K, D, D).contiguous(),
                val.expand(*indep, K, D).reshape(B, K, D).contiguous(), w,
                activity=act, layout=y_layout,
                affiliation_eps=affiliation_eps, want_q=True)
            aff, q = aff.reshape(shape), q.reshape(shape)
            if aligner is not None:
                if device_aligner:
                    aff, q = apply_inline_permutation_alignment(
                        affiliation=aff, quadratic_form=q,
                        weight_constant_axis=weight_constant_axis, aligner=aligner,
                        status_out=aligner_status)
                else:  # a foreign (NumPy) aligner object: the one host excursion left
                    a_h, q_h = apply_inline_permutation_alignment(
                        affiliation=_lib.to_host(aff), quadratic_form=_lib.to_host(q),
                        weight_constant_axis=weight_constant_axis, aligner=aligner)
                    aff = _lib.to_device(a_h, t.float64, device=dev)
                    q = _lib.to_device(q_h, t.float64, device=dev)
            return aff, q

        def m_step(aff, q):
            weight = None
            if weight_hook is not None:
                weight = weight_hook(aff, sal_dev)  # e.g. an all-reduce over the ranks' bins
            elif isinstance(weight_constant_axis, int) and \
                    weight_constant_axis % len(shape) - len(shape) == -2:
                weight = t.full((K, 1), 1.0 / K, dtype=t.float64, device=dev)  # :180-183
            else:
                weight = self._device_weight(aff, sal_dev, weight_constant_axis, indep)
            if weight is None:  # exotic axis sets: the NumPy formula
                host_excursion[0] = True
                weight = _lib.to_device(_host_estimate_mixture_weight(
                    _lib.to_host(aff), None if sal_dev is None else _lib.to_host(sal_dev),
                    weight_constant_axis), t.float64, device=dev)
            masked = aff if sal_dev is None else aff * sal_dev[..., None, :]
            # status words of the M-step: queued like the aligner's, one read-back after the loop
            vec, val, _, st = engine.cacg_m_step(
                yn, masked.reshape(B, K, N).contiguous(), q.reshape(B, K, N).contiguous(),
                layout=y_layout, covariance_norm=covariance_norm,
                eigenvalue_floor=eigenvalue_floor, check_status=False)
            m_status.append(st.reshape(-1))
            return vec.reshape(*indep, K, D, D), val.reshape(*indep, K, D), weight

        # The loop body is a fixed sequence of ~12 launches on device-resident state: after two
        # eager iterations (code objects loaded, allocator warm, every host-side decision taken)
        # ONE iteration is captured into a graph that feeds itself -- its last nodes copy the new
        # model over the inputs of its first -- and replayed for the rest of the fit: one graph
        # launch per EM iteration instead of a dozen kernel launches with their host-side argument
        # marshalling.  Only when nothing in the iteration leaves the device (device aligner or
        # none, mixture weights from the device reduction, no collective hook) and only on request
        # (PBBSS_STEPWISE_GRAPH=1): see _stepwise_graph_enabled for the measurement that keeps it off.
        done = 0
        while done < iterations:
            if vec is not None:
                aff, q = e_step(vec, val, weight)
            vec, val, weight = m_step(aff, q)
            done += 1
            if (done == 2 and iterations - done >= 3 and weight_hook is None
                    and not host_excursion[0] and _stepwise_graph_enabled()):
                left = self._replay_iterations(t, e_step, m_step, vec, val, weight, m_status,
                                               aligner_status, iterations - done)
                if left is not None:
                    vec, val, weight = left
                    done = iterations
        what = 'ComplexAngularCentralGaussianTrainer._fit'
        bits = int(np.bitwise_or.reduce(_lib.to_host(t.cat(m_status)))) if m_status else 0
        # (one-iteration M-step launches never use split groups -- kSplitMinIterations = 3,
        # csrc/em_launch.hpp -- so there is no inter-workgroup time-out to recover from here)
        engine._raise_for_bits(bits, what)
        if aligner_status:
            bits = int(np.bitwise_or.reduce(_lib.to_host(t.cat(aligner_status)).reshape(-1)))
            if bits & _lib.ST_EIG_NOCONV and not bits & _lib.ST_NONFINITE and _retry_team:
                # a wait between the workgroups that share the utterance ran out in one of the
                # iterations (GPU shared with other work): nothing downstream of that mapping
                # is valid -- run the loop again on the one-workgroup kernel
                if weight_hook is not None:
                    # a rank-local repeat of the loop would issue `iterations` collectives its
                    # peers never join (not reachable today: under sharding the aligner is the
                    # synchronous sharded_inline_aligner, which recovers inside the solver)
                    raise RuntimeError('inline DHTV permutation alignment timed out inside a '
                                       'sharded fit; set_dhtv_team(1) avoids the team kernel')
                import warnings
                warnings.warn('inline DHTV permutation alignment: the workgroups of the '
                              'utterance were not co-resident (GPU shared with other work); '
                              'running the fit again on the one-workgroup kernel', RuntimeWarning)
                before = engine.dhtv_team(dev.index)
                engine.set_dhtv_team(1, dev.index)
                try:
                    return self._fit_stepwise(
                        yb, indep, K, gamma0, model, iterations, saliency, sal, act,
                        weight_constant_axis, covariance_norm, affiliation_eps, eigenvalue_floor,
                        hermitize, aligner, like_torch, weight_hook=weight_hook,
                        _retry_team=False)
                finally:
                    engine.set_dhtv_team(before, dev.index)
            if bits != 0:
                raise ValueError('score matrix is infeasible')  # permutation_alignment.py:512-514
        return CACGMM(
            weight=as_result(weight, like_torch),
            cacg=ComplexAngularCentralGaussian(
                covariance_eigenvectors=as_result(vec, like_torch),
                covariance_eigenvalues=as_result(val, like_torch)))

    def fit_with_log_likelihood(self, y, initialization=None, num_classes=None, iterations=100,
                                **fit_kwargs):
        """`fit` that also returns the log-likelihood after every EM iteration (SURVEY section 5
        'metrics': the reference declares a `log_likelihood_history` (gmm.py:31) and never fills
        it; `CACGMM.log_likelihood`, cacgmm.py:97-138, is what it would hold).  The loop runs one
        iteration per launch -- resuming from the fitted model is exact (cacgmm.py:229-234) -- with
        one log-pdf pass after each: a monitoring aid, ~3 launches per iteration instead of one
        fused fit.  Returns (model, history) with history[i] = log-likelihood after iteration i+1."""
        assert iterations > 0, iterations
        like_torch = _lib.is_torch(y)
        yd = _complex_device(y)
        model = self.fit(yd, initialization=initialization, num_classes=num_classes, iterations=1,
                         **fit_kwargs)
        history = [model.log_likelihood(yd)]
        for _ in range(iterations - 1):
            model = self.fit(yd, initialization=model, iterations=1, **fit_kwargs)
            history.append(model.log_likelihood(yd))
        if not like_torch:  # NumPy in, NumPy out
            model = CACGMM(
                weight=as_result(_lib.to_device(model.weight), False),
                cacg=ComplexAngularCentralGaussian(
                    covariance_eigenvectors=as_result(
                        _lib.to_device(model.cacg.covariance_eigenvectors), False),
                    covariance_eigenvalues=as_result(
                        _lib.to_device(model.cacg.covariance_eigenvalues), False)))
        return model, np.asarray(history)

    def fit_predict(
            self,
            y,
            initialization=None,
            num_classes=None,
            iterations=100,
            *,
            saliency=None,
            source_activity_mask=None,
            weight_constant_axis=(-1,),
            hermitize=True,
            covariance_norm='eigenvalue',
            affiliation_eps=1e-10,
            eigenvalue_floor=1e-10,
            inline_permutation_aligner=None,
            _weight_hook=None,
    ):
        """Fit a model, then return the posterior affiliations
        (reference: cacgmm.py:282-313)."""
        model = self.fit(
            y=y, initialization=initialization, num_classes=num_classes,
            iterations=iterations, saliency=saliency,
            source_activity_mask=source_activity_mask,
            weight_constant_axis=weight_constant_axis, hermitize=hermitize,
            covariance_norm=covariance_norm, affiliation_eps=affiliation_eps,
            eigenvalue_floor=eigenvalue_floor,
            inline_permutation_aligner=inline_permutation_aligner,
            _with_affiliation=True, _weight_hook=_weight_hook)
        if isinstance(model, tuple):
            # fused path: the kernel's final E-step IS model.predict(y) (new weights,
            # affiliation_eps = 0, no activity mask) -- no second launch, no re-upload
            return model[1]
        return model.predict(y)
