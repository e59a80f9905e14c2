"""Host-side glue of the mixture-model EM loop used by the STEPWISE path only
(options that couple frequency bins).  The fused kernel implements the same
rules on the device (csrc/cacgmm_em.hpp: finish_sums / phase_e).

Mirrors pb_bss/distribution/mixture_model_utils.py.
"""
import numpy as np

__all__ = [
    'estimate_mixture_weight',
    'log_pdf_to_affiliation',
    'apply_inline_permutation_alignment',
]


def _l1_normalize_where(x, axis, eps):
    """x / sum|x| along axis; a zero sum is replaced by eps
    (reference: distribution/utils.py:223-256 with ord=1, eps_style='where')."""
    s = np.sum(np.abs(x), axis=axis, keepdims=True)
    return x / np.where(s == 0, eps, s)


def estimate_mixture_weight(affiliation, saliency=None, weight_constant_axis=-1):
    """Mixture weights from affiliations (..., K, N).

    Reference: mixture_model_utils.py:133-203.  Plain mean over
    `weight_constant_axis` (kept as singleton); with a saliency (..., N) the
    saliency-weighted sum, L1-normalised over the class axis; an axis that IS
    the class axis yields the constant (K, 1) array 1/K.
    """
    affiliation = np.asarray(affiliation)
    nd = affiliation.ndim
    if isinstance(weight_constant_axis, int) and weight_constant_axis % nd - nd == -2:
        K = affiliation.shape[-2]
        return np.full([K, 1], 1 / K)
    if isinstance(weight_constant_axis, list):
        weight_constant_axis = tuple(weight_constant_axis)
    if saliency is None:
        return affiliation.mean(axis=weight_constant_axis, keepdims=True)
    weighted = (affiliation * saliency[..., None, :]).sum(
        axis=weight_constant_axis, keepdims=True)
    return _l1_normalize_where(weighted, axis=-2, eps=1e-10)


def log_pdf_to_affiliation(weight, log_pdf, source_activity_mask=None,
                           affiliation_eps=0.):
    """Posterior from class log-pdfs (..., K, N): max-shifted exp, times weight
    (and activity mask), normalised with a `tiny` floor, clipped to
    [eps, 1-eps] without re-normalisation.  Reference:
    mixture_model_utils.py:7-55.  (Host version for callers outside the fused
    kernel, e.g. models that add a second log-pdf before the softmax.)"""
    shifted = log_pdf - log_pdf.max(axis=-2, keepdims=True)
    post = np.exp(shifted) * weight
    if source_activity_mask is not None:
        assert source_activity_mask.dtype == bool, source_activity_mask.dtype
        post = post * source_activity_mask
    post = post / np.maximum(post.sum(axis=-2, keepdims=True),
                             np.finfo(post.dtype).tiny)
    if affiliation_eps != 0:
        post = np.clip(post, affiliation_eps, 1 - affiliation_eps)
    return post


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
