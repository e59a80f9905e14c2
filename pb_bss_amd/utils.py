"""The handful of array helpers from pb_bss/utils.py that the hot-path modules re-export
(`from pb_bss.utils import ...` in gmm.py, gcacgmm.py, beamformer_wrapper.py, the distribution
classes): shape bookkeeping on the host, no arithmetic.  Reference: pb_bss/utils.py:185-345."""
import numpy as np

__all__ = ['is_broadcast_compatible', 'labels_to_one_hot', 'unsqueeze', 'get_pca']


def is_broadcast_compatible(*shapes):
    """True if the shapes broadcast against each other (utils.py:185-194): aligned from the
    right, every axis holds at most one size other than 1."""
    longest = max((len(s) for s in shapes), default=0)
    for pos in range(1, longest + 1):
        sizes = {s[-pos] for s in shapes if len(s) >= pos} - {1}
        if len(sizes) > 1:
            return False
    return True


def labels_to_one_hot(labels, categories, axis=0, keepdims=False, dtype=bool):
    """Integer labels of any shape -> one-hot array with the `categories` axis at `axis`
    (utils.py:197-279).  keepdims=True replaces an existing singleton axis instead of
    inserting a new one."""
    labels = np.asarray(labels)
    if keepdims:
        assert labels.shape[axis] == 1, (labels.shape, axis)
        labels = np.squeeze(labels, axis=axis)
    assert np.issubdtype(labels.dtype, np.integer), labels.dtype
    assert labels.size == 0 or (labels.min() >= 0 and labels.max() < categories), \
        (labels.min(), labels.max(), categories)
    hot = labels[..., None] == np.arange(categories)         # categories last
    axis = axis % hot.ndim
    return np.moveaxis(hot, -1, axis).astype(dtype)


def unsqueeze(array, axis):
    """Insert singleton axes at the (final-array) positions `axis` (utils.py:306-345):
    unsqueeze(ones((2, 3)), (-3, -1)).shape == (2, 1, 3, 1)."""
    array = np.array(array)
    axis = (axis,) if isinstance(axis, int) else tuple(axis)
    nd = array.ndim + len(axis)
    where = []
    for a in axis:
        if not -nd <= a < nd:
            raise IndexError(np.shape(array), list(array.shape), axis)
        where.append(a % nd)
    shape = list(array.shape)
    for p in sorted(where):
        shape.insert(p, 1)
    return array.reshape(shape)


def get_pca(target_psd_matrix, return_all_vecs=False):
    """Dominant eigenpair (or all of them) of Hermitian matrices on the device -- the reference
    keeps this one in pb_bss/utils.py and re-exports it from extraction/beamformer.py; here the
    device function lives in pb_bss_amd.extraction.beamformer (`pbbss_heev_batched`)."""
    from .extraction.beamformer import get_pca as device_get_pca
    return device_get_pca(target_psd_matrix, return_all_vecs=return_all_vecs)
