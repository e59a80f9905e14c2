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

__all__ = ['shard_bounds', 'shard_sizes', 'all_gather_bins', 'all_reduce_sum', 'fit_predict_sharded',
           'sharded_inline_aligner',
           'fit_predict_sharded_joint', 'native_comm',
           'init_native_comm', 'destroy_native_comm']

# device index -> (world, rank, group) of the RCCL communicator created in the library handle;
# `group` is the torch.distributed group it was built for (None = the world): all_gather_bins
# takes the native path only for calls on that very group
_NATIVE_COMM = {}


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


def init_native_comm(group=None, device_index=None):
    """Create the library's own RCCL communicator for this rank (C ABI `pbbss_comm_create`,
    csrc/comm.hip): rank 0 draws the 128-byte unique id, torch.distributed (any backend) carries
    it to the other ranks -- the only thing the Python host contributes; a C++ / Go / Java host
    would use MPI or a socket.  Afterwards `all_gather_bins` on CUDA tensors goes through
    `pbbss_allgather_masks` instead of torch.distributed."""
    import ctypes
    import torch
    import torch.distributed as dist
    from . import _lib
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index in _NATIVE_COMM:
        raise RuntimeError('this device already holds a library communicator (one per handle): '
                           'call destroy_native_comm() first')
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    uid = ctypes.create_string_buffer(128)
    if rank == 0:
        _lib.check(_lib.load().pbbss_comm_unique_id(uid), 'comm_unique_id')
    box = [bytes(uid.raw)]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                               group=group)
    _lib.check(_lib.load().pbbss_comm_create(_lib.handle(device_index), box[0], world, rank),
               'comm_create')
    _NATIVE_COMM[device_index] = (world, rank, group)


def native_comm(device_index=None):
    """(world, rank, group) of the library communicator of this device, or None."""
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device()
    return _NATIVE_COMM.get(device_index)


def destroy_native_comm(device_index=None):
    import torch
    from . import _lib
    if device_index is None:
        device_index = torch.cuda.current_device()
    if _NATIVE_COMM.pop(device_index, None) is not None:
        _lib.check(_lib.load().pbbss_comm_destroy(_lib.handle(device_index)), 'comm_destroy')


def _native_all_gather(local, num_bins, bin_axis):
    """`pbbss_allgather_masks`: (outer, n_local, inner) -> (outer, num_bins, inner), any 4- or
    8-byte element type (the kernels only move bits), complex128 as pairs of float64."""
    import torch
    from . import _lib
    x = local.contiguous()
    if x.dtype == torch.complex128:
        full = _native_all_gather(torch.view_as_real(x), num_bins, bin_axis)
        return torch.view_as_complex(full.contiguous())
    shape = list(x.shape)
    outer = int(np.prod(shape[:bin_axis], dtype=np.int64))
    inner = int(np.prod(shape[bin_axis + 1:], dtype=np.int64))
    out_shape = shape[:bin_axis] + [num_bins] + shape[bin_axis + 1:]
    out = torch.empty(out_shape, dtype=x.dtype, device=x.device)
    rc = _lib.load().pbbss_allgather_masks(
        _lib.handle(x.device.index), _lib.ptr(x) if x.numel() else None, x.element_size(), outer,
        num_bins, inner, _lib.ptr(out), _lib.stream_ptr(x.device.index))
    _lib.check(rc, f'allgather_masks(outer={outer}, bins={num_bins}, inner={inner})')
    return out


def all_gather_bins(local, num_bins, bin_axis=0, group=None):
    """All-gather a tensor sharded along `bin_axis` into the full (num_bins, ...)
    tensor on every rank.  `local` holds this rank's block (shard_bounds).
    CUDA tensors go through the library's own RCCL communicator when `init_native_comm` has
    created one (C ABI `pbbss_allgather_masks`), otherwise -- and for CPU tensors (gloo) --
    through torch.distributed; both pad to the largest block, gather once and trim."""
    import torch
    import torch.distributed as dist
    bin_axis = bin_axis % local.ndim
    if (local.is_cuda and local.device.index in _NATIVE_COMM
            and _NATIVE_COMM[local.device.index][2] is group
            and local.element_size() in (4, 8, 16) and
            (local.element_size() != 16 or local.dtype == torch.complex128)):
        world, rank, _ = _NATIVE_COMM[local.device.index]
        assert local.shape[bin_axis] == shard_sizes(num_bins, world)[rank], (
            local.shape, shard_sizes(num_bins, world), rank)
        return _native_all_gather(local, num_bins, bin_axis)
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


def shared_weight_allreduce(weight_constant_axis, total_bins, bin_axis=-3, group=None):
    """Hook for `CACGMMTrainer.fit(_weight_hook=...)` when `weight_constant_axis` contains the
    SHARDED bin axis: estimate_mixture_weight (mixture_model_utils.py:133-203) over the bins of
    all ranks -- the local (saliency-masked) sums, ONE small all-reduce per EM iteration
    (K x T float64 for weight_constant_axis=(-3,)), then the reference's normalisation with the
    GLOBAL bin count.  Works on whatever device the affiliations live on (device tensors with
    nccl / RCCL, CPU tensors with gloo)."""
    import torch
    import torch.distributed as dist
    axes_in = (weight_constant_axis,) if isinstance(weight_constant_axis, int) \
        else tuple(weight_constant_axis)

    def hook(aff, sal):
        nd = aff.ndim
        axes = tuple(sorted({a % nd for a in axes_in}))
        assert (nd - 2) not in axes, 'the class axis cannot be averaged with others'
        b_ax = bin_axis % nd
        assert b_ax in axes, (weight_constant_axis, bin_axis)
        masked = aff if sal is None else aff * sal[..., None, :]
        s = masked.sum(dim=axes, keepdim=True)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(s, group=group)
        if sal is None:                                  # :186-188: a mean
            count = 1
            for a in axes:
                count *= total_bins if a == b_ax else aff.shape[a]
            return s / count
        norm = s.abs().sum(dim=-2, keepdim=True)         # :190-201: L1 over the classes
        return s / torch.where(norm == 0, torch.full_like(norm, 1e-10), norm)

    def idle(shape, dtype, device, iterations):
        """A rank without bins still takes part in every iteration's all-reduce."""
        if dist.is_available() and dist.is_initialized():
            for _ in range(iterations):
                dist.all_reduce(torch.zeros(shape, dtype=dtype, device=device), group=group)

    hook.idle = idle
    return hook


def all_reduce_sum(x, group=None):
    """Sum a small tensor over the ranks (in place, returned).  Complex tensors travel as
    (re, im) pairs; without an initialised process group (single process) this is the
    identity.  Used for the cross-bin reductions of SURVEY section 8e: the D-vector of the
    MVDR-Souden reference-channel SNR (beamformer.py:601-624) and the shared mixture weights."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return x
    if x.is_complex():
        r = torch.view_as_real(x.contiguous()).contiguous()
        dist.all_reduce(r, group=group)
        return torch.view_as_complex(r)
    dist.all_reduce(x, group=group)
    return x


class sharded_inline_aligner:
    """An `inline_permutation_aligner` under bin sharding.  The reference hands the solver the
    affiliations of ALL bins after every E-step (cacgmm.py:260-267, mixture_model_utils.py:264-306):
    `calculate_mapping` all-gathers the ranks' (K, F_local, T) blocks, every rank solves the full
    (K, F) mapping from bit-identical inputs (so all ranks hold the same mapping without a second
    collective) and keeps the columns of its own block; `apply_mapping` then acts on the local
    block.  One mask all-gather per EM iteration -- the same traffic as the final gather of
    `fit_predict_sharded`.  Wraps the library's device aligners (tensors stay on the device) as
    well as foreign NumPy aligner objects (one host excursion per iteration, as unsharded)."""

    def __init__(self, aligner, total_bins, group=None):
        import torch.distributed as dist
        self.aligner = aligner
        self.F = total_bins
        self.group = group
        self.lo, self.hi = shard_bounds(total_bins, dist.get_world_size(group), dist.get_rank(group))
        self._device = type(aligner).__module__.startswith('pb_bss_amd')

    def calculate_mapping(self, kft):
        from . import _lib
        t = _lib.torch()
        is_tensor = isinstance(kft, t.Tensor)
        local = kft if is_tensor else t.from_numpy(np.ascontiguousarray(kft))
        full = all_gather_bins(local.contiguous(), self.F, bin_axis=1, group=self.group)
        if self._device:
            mapping = self.aligner.calculate_mapping(full)
        else:
            mapping = self.aligner.calculate_mapping(
                full.cpu().numpy() if is_tensor else full.numpy())
        block = mapping[:, self.lo:self.hi]
        if isinstance(block, t.Tensor):
            return block.contiguous()
        block = np.ascontiguousarray(block)
        return t.from_numpy(block).to(kft.device) if is_tensor else block

    def apply_mapping(self, x, mapping):
        from . import _lib
        t = _lib.torch()
        if self._device or not isinstance(x, t.Tensor):
            return self.aligner.apply_mapping(x, mapping)
        out = self.aligner.apply_mapping(x.cpu().numpy(), mapping.cpu().numpy())
        return t.from_numpy(np.ascontiguousarray(out)).to(x.device)

    def idle(self, K, T, dtype, device, count):
        """A rank without bins keeps the collective schedule: `count` gathers of an empty block."""
        from . import _lib
        t = _lib.torch()
        for _ in range(count):
            all_gather_bins(t.empty((K, 0, T), dtype=dtype, device=device), self.F, bin_axis=1,
                            group=self.group)


def _bin_block(x, axis_from_end, lo, hi):
    """Slice bins [lo, hi) out of an array whose bin axis is `axis_from_end` (negative);
    arrays that are absent or broadcast along the bins pass through."""
    if x is None or x.ndim < -axis_from_end or x.shape[axis_from_end] == 1:
        return x
    sl = [slice(None)] * x.ndim
    sl[axis_from_end] = slice(lo, hi)
    return x[tuple(sl)]


def fit_predict_sharded(y, initialization, iterations=100, *, trainer=None, bin_axis=-3,
                        group=None, **fit_kwargs):
    """Sharded `fit_predict` of a per-bin mixture trainer: every rank passes the FULL problem
    description (y (..., F, T, D), initialization (..., F, K, T)); each fits only its own block
    of frequency bins and the masks are all-gathered, so every rank returns the complete
    (..., F, K, T) affiliations, ready for permutation alignment.

    `trainer`: an instance (or class) of `CACGMMTrainer` (default), `CWMMTrainer` (BASELINE
    configs[3], reference cwmm.py:76-149: every bin is an independent Watson mixture) or
    `VMFMMTrainer` / `GMMTrainer` (independent leading axes, vmfmm.py:124-172).  The joint
    spatial + spectral trainers couple the bins through ONE spectral mixture: see
    `fit_predict_sharded_joint`.

    `weight_constant_axis` may contain the sharded bin axis ((-3,), (-3, -1): weights
    averaged over the bins of ALL ranks; CACGMMTrainer only): the fit then runs step by step
    with one small all-reduce per iteration (`shared_weight_allreduce`).
    `inline_permutation_aligner` (needs such bin-constant weights, as in the reference): the
    solver is wrapped in `sharded_inline_aligner` -- one mask all-gather per EM iteration, every
    rank solves the full mapping and applies the columns of its own block (CACGMMTrainer, 3-D y).
    """
    import torch.distributed as dist
    from . import _lib
    from .distribution import CACGMMTrainer
    if trainer is None:
        trainer = CACGMMTrainer()
    elif isinstance(trainer, type):
        trainer = trainer()
    is_cacgmm = isinstance(trainer, CACGMMTrainer)
    aligner = fit_kwargs.get('inline_permutation_aligner')
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nd_y = y.ndim
    f_axis_y = bin_axis % nd_y
    assert f_axis_y <= nd_y - 3, (bin_axis, tuple(y.shape), 'bin_axis must be an independent axis')
    F = y.shape[f_axis_y]
    # weights shared across the sharded axis would need an all-reduce per EM iteration
    # (mixture_model_utils.py:133-203): with y (..., F, T, D) the affiliation is (..., F, K, T),
    # i.e. the bin axis sits at f_axis_y - nd_y (same negative index in both arrays)
    wca = fit_kwargs.get('weight_constant_axis', (-1,))
    wca = (wca,) if isinstance(wca, int) else tuple(wca)
    coupled = any(a % nd_y == f_axis_y for a in wca)
    hook = None
    if coupled:
        if not is_cacgmm:
            raise NotImplementedError(
                f'{type(trainer).__name__}: mixture weights shared over the sharded axis need a '
                'collective inside the EM loop, which only CACGMMTrainer provides')
        assert not (len(wca) == 1 and wca[0] % nd_y - nd_y == -2), wca
        hook = shared_weight_allreduce(wca, F, bin_axis=f_axis_y - nd_y, group=group)
    if aligner is not None:
        # mixture_model_utils.py:264-306: 3-D affiliations and frequency-constant weights
        assert is_cacgmm and nd_y == 3 and coupled, (
            'inline_permutation_aligner under bin sharding: CACGMMTrainer, y (F, T, D) and a '
            f'weight_constant_axis that contains the bin axis (got {wca}, y.ndim = {nd_y})')
        aligner = sharded_inline_aligner(aligner, F, group=group)
    lo, hi = shard_bounds(F, world, rank)
    neg = f_axis_y - nd_y               # bin axis of y (..., F, T, D) and of (..., F, K, T)
    y_loc = _bin_block(y, neg, lo, hi)
    i_loc = _bin_block(initialization, neg, lo, hi)
    kwargs = dict(fit_kwargs)
    if kwargs.get('saliency') is not None:              # (..., F, T): one axis fewer
        kwargs['saliency'] = _bin_block(kwargs['saliency'], neg + 1, lo, hi)
    if kwargs.get('source_activity_mask') is not None:  # (..., F, K, T)
        kwargs['source_activity_mask'] = _bin_block(kwargs['source_activity_mask'], neg, lo, hi)
    K = initialization.shape[-2]
    if hook is not None:
        kwargs['_weight_hook'] = hook
    if aligner is not None:
        kwargs['inline_permutation_aligner'] = aligner
    # the library's trainers take device tensors; a CPU stand-in (the gloo tests run this very
    # function with oracle-backed trainers) says so with a `_to_device` of its own
    prep = getattr(trainer, '_to_device', _lib.to_device)
    t = _lib.torch()
    if hi > lo:
        masks = trainer.fit_predict(prep(y_loc), initialization=prep(i_loc),
                                    iterations=iterations, **kwargs)
    else:
        dev = y.device if (isinstance(y, t.Tensor) and not hasattr(trainer, '_to_device')
                           and y.is_cuda) else (
            t.device('cpu') if hasattr(trainer, '_to_device')
            else t.device('cuda', t.cuda.current_device()))
        # keep the collective schedule of the ranks that own bins, IN THEIR ORDER (the step-wise
        # loop of CACGMMTrainer, cacgmm.py:252-278): per EM iteration first -- from the second
        # iteration on -- the aligner's mask gather after the E-step, then the weight hook's
        # all-reduce before the M-step.  Issuing all all-reduces first and the gathers afterwards
        # would pair this rank's n-th collective with a different one on its peers.
        wshape = None
        if hook is not None:
            nd_a = initialization.ndim
            red = sorted({a % nd_a for a in wca})
            wshape = [1 if ax in red else n for ax, n in enumerate(initialization.shape)]
        for it in range(iterations):
            if aligner is not None and it > 0:
                aligner.idle(K, y.shape[-2], t.float64, dev, 1)
            if hook is not None:
                hook.idle(wshape, t.float64, dev, 1)
        # more ranks than bins: this rank owns nothing and contributes an empty block
        shape = list(y.shape[:-2]) + [K, y.shape[-2]]
        shape[f_axis_y] = 0
        masks = t.empty(shape, dtype=t.float64, device=dev)
    return all_gather_bins(masks, F, bin_axis=f_axis_y, group=group)


def fit_predict_sharded_joint(trainer, observation, embedding, initialization, iterations=100, *,
                              group=None, **fit_kwargs):
    """Sharded `fit_predict` of the joint spatial + spectral trainers (`GCACGMMTrainer`,
    `VMFCACGMMTrainer`; BASELINE configs[4], reference gcacgmm.py:121-225, vmfcacgmm.py:34-301).

    observation (F, T, D), embedding (F, T, E), initialization (F, K, T): every rank passes the
    full arrays and fits its own block of frequency bins.  The cACG half is per bin; the spectral
    half is ONE mixture over all F*T points, so its M-step sums (K x (E + 1) numbers per chunk of
    points, gaussian.py:152-193 / von_mises_fisher.py:122-144) are summed over the ranks once per
    EM iteration -- an `ncclAllReduce` the LIBRARY enqueues on the caller's stream between the
    partial-sum kernel and the finalize kernel (C ABI `pbbss_mix_opts.sharded`, communicator of
    `init_native_comm`): no host round trip inside the loop, and every rank ends up with
    bit-identical spectral parameters.  Mixture weights that are constant over the bins
    ((-3, -1), (-3,)) are reduced the same way.  Returns the gathered (F, K, T) affiliations.
    """
    import torch.distributed as dist
    from . import _lib
    if isinstance(trainer, type):
        trainer = trainer()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = _lib.torch()
    dev = t.cuda.current_device()
    if world > 1:
        comm = native_comm(dev)
        assert comm is not None and comm[2] is group and comm[0] == world, (
            'fit_predict_sharded_joint needs the library communicator of this group: call '
            'sharding.init_native_comm(group) first')
    F = observation.shape[0]
    lo, hi = shard_bounds(F, world, rank)
    assert hi > lo, ('a rank without bins cannot take part in the in-library all-reduce', F, world)
    kwargs = dict(fit_kwargs)
    if kwargs.get('saliency') is not None:
        kwargs['saliency'] = _bin_block(kwargs['saliency'], -2, lo, hi)
    from .distribution import _joint
    with _joint.sharded_bins(world > 1):
        masks = trainer.fit_predict(
            _lib.to_device(observation[lo:hi]), _lib.to_device(embedding[lo:hi]),
            initialization=_lib.to_device(initialization[lo:hi]), iterations=iterations, **kwargs)
    return all_gather_bins(masks, F, bin_axis=0, group=group)
