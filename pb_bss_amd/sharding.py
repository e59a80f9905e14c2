"""Frequency-bin sharding of the cACGMM EM across the GPUs of one node.

Every frequency bin is an independent EM problem under the default
weight_constant_axis=(-1,) (reference: distribution/cacgmm.py:151, :204), so
bins shard over ranks with NO collective inside the EM loop.  The only
exchange is one all-gather of the posterior masks (F_local, K, T) -> (F, K, T)
before permutation alignment, which needs all bins of an utterance together
(reference: permutation_alignment.py:334).  One process per GPU,
torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests of the gather logic.

F is generally not divisible by the world size (513 = 8*64 + 1): shards are
contiguous blocks whose sizes differ by at most one; the gather pads every
shard to the largest block and trims after the collective.

The same helpers shard ANY independent axis: with whole utterances per rank
(`bin_axis` = the utterance axis) nothing is exchanged until the final gather
(examples/separate_batch.py --shard utterances), which is preferable once
there are at least as many utterances as GPUs.
"""
import numpy as np

__all__ = ['shard_bounds', 'shard_sizes', 'all_gather_bins', 'fit_predict_sharded']


def shard_sizes(num_bins, world_size):
    """Sizes of the contiguous blocks: the first (num_bins % world) ranks get one
    extra bin."""
    base, extra = divmod(int(num_bins), int(world_size))
    return [base + (1 if r < extra else 0) for r in range(world_size)]


def shard_bounds(num_bins, world_size, rank):
    """[start, stop) of the bins owned by `rank`."""
    sizes = shard_sizes(num_bins, world_size)
    start = int(np.sum(sizes[:rank]))
    return start, start + sizes[rank]


def all_gather_bins(local, num_bins, bin_axis=0, group=None):
    """All-gather a tensor sharded along `bin_axis` into the full (num_bins, ...)
    tensor on every rank.  `local` holds this rank's block (shard_bounds).
    Works for CUDA tensors (RCCL) and CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(num_bins, world)
    assert local.shape[bin_axis] == sizes[rank], (local.shape, sizes, rank)
    x = local.movedim(bin_axis, 0).contiguous()
    pad_to = max(sizes)
    if x.shape[0] < pad_to:
        pad = torch.zeros((pad_to - x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        x = torch.cat([x, pad], dim=0)
    out = torch.empty((world * pad_to, *x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    pieces = [out[r * pad_to:r * pad_to + sizes[r]] for r in range(world)]
    full = torch.cat(pieces, dim=0)
    return full.movedim(0, bin_axis)


def fit_predict_sharded(y, initialization, iterations=100, *, bin_axis=-3,
                        group=None, **fit_kwargs):
    """Sharded `CACGMMTrainer.fit_predict`: every rank passes the FULL problem
    description (y (..., F, T, D), initialization (..., F, K, T)); each fits
    only its own block of frequency bins and the masks are all-gathered, so
    every rank returns the complete (..., F, K, T) affiliations, ready for
    permutation alignment.

    Only options that keep bins independent are allowed (no -3 in
    weight_constant_axis, no inline_permutation_aligner).
    """
    import torch.distributed as dist
    from . import _lib
    from .distribution import CACGMMTrainer
    assert 'inline_permutation_aligner' not in fit_kwargs or \
        fit_kwargs['inline_permutation_aligner'] is None
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nd_y = y.ndim
    f_axis_y = bin_axis % nd_y
    F = y.shape[f_axis_y]
    lo, hi = shard_bounds(F, world, rank)
    sl_y = [slice(None)] * nd_y
    sl_y[f_axis_y] = slice(lo, hi)
    sl_i = [slice(None)] * initialization.ndim
    sl_i[bin_axis % initialization.ndim] = slice(lo, hi)
    y_loc = _lib.to_device(y[tuple(sl_y)])
    i_loc = _lib.to_device(initialization[tuple(sl_i)])
    masks = CACGMMTrainer().fit_predict(y_loc, initialization=i_loc,
                                        iterations=iterations, **fit_kwargs)
    return all_gather_bins(masks, F, bin_axis=bin_axis % masks.ndim, group=group)
