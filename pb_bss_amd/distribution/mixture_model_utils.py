"""The public helpers of pb_bss/distribution/mixture_model_utils.py as stand-alone device steps
(NumPy in -> NumPy out, device tensors in -> device tensors out).  The fused kernels implement the
same rules inside their loops (csrc/cacgmm_em.hpp: finish_sums / phase_e); the step-wise trainers
call these.
"""
import numpy as np

from .. import _lib

__all__ = [
    'estimate_mixture_weight',
    'log_pdf_to_affiliation',
    'log_pdf_to_affiliation_for_integration_models_with_inline_pa',
    'apply_inline_permutation_alignment',
]


def _l1_normalize_where(x, axis, eps):
    """x / sum|x| along axis; a zero sum is replaced by eps
    (reference: distribution/utils.py:223-256 with ord=1, eps_style='where')."""
    s = np.sum(np.abs(x), axis=axis, keepdims=True)
    return x / np.where(s == 0, eps, s)


def _host_estimate_mixture_weight(affiliation, saliency, weight_constant_axis):
    """The formula itself, for the axis sets pbbss_estimate_mixture_weight does not serve (a tuple
    that contains the class axis, a non-trailing block of independent axes, a saliency with more
    than 16 classes): a mean / normalised sum over a handful of axes, not on the hot path."""
    if saliency is None:
        return affiliation.mean(axis=weight_constant_axis, keepdims=True)
    weighted = (affiliation * saliency[..., None, :]).sum(
        axis=weight_constant_axis, keepdims=True)
    return _l1_normalize_where(weighted, axis=-2, eps=1e-10)


def estimate_mixture_weight(affiliation, saliency=None, weight_constant_axis=-1):
    """Mixture weights from affiliations (..., K, N).

    Reference: mixture_model_utils.py:133-203.  Plain mean over
    `weight_constant_axis` (kept as singleton); with a saliency (..., N) the
    saliency-weighted sum, L1-normalised over the class axis; an axis that IS
    the class axis yields the constant (K, 1) array 1/K.  Runs
    `pbbss_estimate_mixture_weight` (csrc/mixw.hip) for the frame axis and / or a
    trailing block of independent axes -- every axis set the trainers produce.
    """
    like_torch = _lib.is_torch(affiliation)
    nd = affiliation.ndim
    if isinstance(weight_constant_axis, int) and weight_constant_axis % nd - nd == -2:
        K = affiliation.shape[-2]
        if like_torch:
            t = _lib.torch()
            return t.full((K, 1), 1 / K, dtype=t.float64, device=affiliation.device)
        return np.full([K, 1], 1 / K)
    if isinstance(weight_constant_axis, list):
        weight_constant_axis = tuple(weight_constant_axis)
    from .cacgmm import CACGMMTrainer
    t = _lib.torch()
    aff = _lib.to_device(affiliation, t.float64)
    sal = None if saliency is None else _lib.to_device(saliency, t.float64)
    w = CACGMMTrainer._device_weight(aff, sal, weight_constant_axis, tuple(aff.shape[:-2]))
    if w is not None:
        return w if like_torch else _lib.to_host(w)
    w = _host_estimate_mixture_weight(
        np.asarray(_lib.to_host(aff)), None if sal is None else _lib.to_host(sal),
        weight_constant_axis)
    return _lib.to_device(w, t.float64) if like_torch else w


def _flat3(x, shape, dtype):
    """Broadcast x against `shape` = (*lead, K, N) and flatten the leading axes -> (B, K', N')
    device tensor in which an axis that did not vary stays a singleton (zero stride downstream)."""
    t = _lib.torch()
    x = _lib.to_device(x, dtype)
    while x.ndim < len(shape):
        x = x.unsqueeze(0)
    if any(a != 1 for a in x.shape[:-2]):
        x = x.expand(*shape[:-2], *x.shape[-2:])
    return x.reshape(-1, *x.shape[-2:]).contiguous()


def log_pdf_to_affiliation(weight, log_pdf, source_activity_mask=None,
                           affiliation_eps=0.):
    """Posterior from class log-pdfs (..., K, N): max-shifted exp, times weight
    (and activity mask), normalised with a `tiny` floor, clipped to
    [eps, 1-eps] without re-normalisation.  Reference:
    mixture_model_utils.py:7-55 -- `pbbss_log_pdf_to_affiliation`."""
    from .. import engine
    t = _lib.torch()
    like_torch = _lib.is_torch(log_pdf)
    shape = tuple(np.broadcast_shapes(
        tuple(np.shape(weight)), tuple(log_pdf.shape),
        *(() if source_activity_mask is None else (tuple(source_activity_mask.shape),))))
    lp = _flat3(log_pdf, shape, t.float64)
    if lp.shape[-2:] != shape[-2:] or lp.shape[0] != int(np.prod(shape[:-2], dtype=np.int64)):
        lp = lp.expand(int(np.prod(shape[:-2], dtype=np.int64)), *shape[-2:]).contiguous()
    act = None
    if source_activity_mask is not None:
        dt = source_activity_mask.dtype
        assert dt in (bool, np.bool_, t.bool), dt  # mixture_model_utils.py:40
        act = _flat3(source_activity_mask, shape, t.uint8).expand(lp.shape).contiguous()
    out = engine.log_pdf_to_affiliation(lp, _flat3(weight, shape, t.float64), act, affiliation_eps)
    out = out.reshape(shape)
    return out if like_torch else _lib.to_host(out)


def log_pdf_to_affiliation_for_integration_models_with_inline_pa(
        weight, spatial_log_pdf, spectral_log_pdf, source_activity_mask=None,
        affiliation_eps=0.):
    """Inline permutation alignment of the integration models (mixture_model_utils.py:58-130):
    per frequency bin the class permutation of the spatial log-pdf that agrees best with the
    spectral one -- `sum_{k,t} softmax_k(lp) lp` over all K! permutations in
    itertools.permutations order, the first strict maximum wins -- then `log_pdf_to_affiliation`
    of `spatial[f, perm] + spectral[f]`.  Both log-pdfs (F, K, T); `weight` broadcastable;
    `pbbss_log_pdf_to_affiliation_inline_pa` (one workgroup per bin), K <= 6."""
    from .. import engine
    t = _lib.torch()
    like_torch = _lib.is_torch(spatial_log_pdf)
    F, K, T = spatial_log_pdf.shape
    act = None
    if source_activity_mask is not None:
        act = _lib.to_device(source_activity_mask, t.uint8).expand(F, K, T).contiguous()
    out = engine.log_pdf_to_affiliation_inline_pa(
        _lib.to_device(spatial_log_pdf, t.float64), _lib.to_device(spectral_log_pdf, t.float64),
        _lib.to_device(weight, t.float64), act, affiliation_eps)
    return out if like_torch else _lib.to_host(out)


def apply_inline_permutation_alignment(affiliation, *, quadratic_form=None,
                                       weight_constant_axis, aligner, status_out=None):
    """Run a permutation-alignment solver between E- and M-step.

    Reference: mixture_model_utils.py:264-306.  affiliation / quadratic_form
    are (F, K, T); `aligner` is any object with
    calculate_mapping((K, F, T)) -> (K, F) and apply_mapping(x, mapping)
    (e.g. pb_bss.permutation_alignment.DHTVPermutationAlignment).

    `status_out` (a list; not in the reference) opts a device caller into the asynchronous
    route: with device tensors and an aligner that offers `calculate_mapping_async` the
    solver's status words are appended to the list instead of being read back here, and the
    caller checks them when it next synchronises.
    """
    msg = ('Inline permutation alignment needs affiliation.ndim == 3 '
           f'({affiliation.shape}) and a frequency-constant mixture weight '
           f'(weight_constant_axis={weight_constant_axis}).')
    assert affiliation.ndim == 3, msg
    assert weight_constant_axis in ((-3,), (-3, -1), -3), msg
    def swap(x):  # (F, K, T) <-> (K, F, T) for NumPy arrays and torch tensors alike
        return x.permute(1, 0, 2).contiguous() if hasattr(x, 'permute') else x.transpose(1, 0, 2)

    if status_out is not None and hasattr(affiliation, 'permute') and affiliation.is_cuda \
            and hasattr(aligner, 'calculate_mapping_async'):
        # device loop (CACGMMTrainer._fit_stepwise): no host synchronisation per EM iteration --
        # the status words are queued for the caller -- and the reverse mapping is applied as a
        # gather along the class axis of the (F, K, T) arrays themselves, without the two
        # transposed copies per array of the generic route below
        import torch as t
        F, K, T = affiliation.shape
        mapping, st = aligner.calculate_mapping_async(
            affiliation.to(t.float64).permute(1, 0, 2).contiguous()[None])
        status_out.append(st)
        idx = mapping[0].t().to(t.int64)[:, :, None].expand(F, K, T)
        aligned = affiliation.gather(1, idx)
        if quadratic_form is None:
            return aligned
        return aligned, quadratic_form.gather(1, idx)
    kft = swap(affiliation)
    mapping = aligner.calculate_mapping(kft)
    aligned = swap(aligner.apply_mapping(kft, mapping))
    if quadratic_form is None:
        return aligned
    q = aligner.apply_mapping(swap(quadratic_form), mapping)
    return aligned, swap(q)
