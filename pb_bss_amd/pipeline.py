"""Mask-based separation of a batch of utterances (the shape of BASELINE config 3).

    STFT batch (U, F, T, D)
      --cACGMM EM--------------------------------->  masks (U, F, K, T)
      --[RCCL all-gather of the masks when the bins are sharded]
      --DHTV permutation alignment---------------->  mapping (U, K, F), aligned masks (U, K, F, T)
      --PSD per class, 'gev+ban' beamformer-------->  w (U, K, F, D)
      --apply------------------------------------->  enhanced (U, K, F, T)

The chain is the reference's canonical recipe (examples/mixture_model_example.ipynb cells
11-12, tests/test_distribution/test_spatial_mm.py:43-49: `CACGMMTrainer.fit` -> `predict` ->
`DHTVPermutationAlignment.from_stft_size(...)(...)` -> `get_power_spectral_density_matrix` ->
`get_bf_vector('gev+ban', target, noise)` -> `apply_beamforming_vector`), stated once here so
that the benchmark, the example and the tests run the same code.

Multi-GPU (one process per GPU, torch.distributed):

* ``shard='bins'``: every rank owns a contiguous block of frequency bins of EVERY utterance
  (the EM needs no collective); the masks are all-gathered once (float32 on request: half
  the xGMI bytes), each rank computes the DHTV mapping of its share of the utterances and
  the tiny (U, K, F) mappings are all-gathered; alignment, PSD, beamformer and apply then run
  on the rank's own bins.  The enhanced signals stay sharded by bins unless ``gather_output``.
* ``shard='utterances'``: whole utterances per rank, no collective until the optional final
  gather (preferable once there are at least as many utterances as GPUs).

The device stages are passed in through ``ops`` so that the CPU tests can run this very
orchestration (slicing, gathers, trimming) under gloo with NumPy stand-ins for the kernels.
"""
import numpy as np

from .sharding import all_gather_bins, shard_bounds

__all__ = ['separate', 'device_ops', 'graphed']


class device_ops:
    """The device implementation of the five stages (torch CUDA tensors in and out)."""

    @staticmethod
    def em_masks(Y, init, iterations):
        from .distribution import CACGMMTrainer
        return CACGMMTrainer().fit_predict(Y, initialization=init, iterations=iterations)

    @staticmethod
    def dhtv_mapping(mask_kft, stft_size):
        from .permutation_alignment import DHTVPermutationAlignment
        return DHTVPermutationAlignment.from_stft_size(stft_size).calculate_mapping(mask_kft)

    @staticmethod
    def apply_mapping(mask_kft, mapping):
        from .permutation_alignment import apply_mapping
        return apply_mapping(mask_kft, mapping)

    @staticmethod
    def psd(X, mask_fkt):
        from .extraction import get_power_spectral_density_matrix
        return get_power_spectral_density_matrix(X, mask_fkt)

    @staticmethod
    def gev_ban(target, noise):
        from .extraction import get_bf_vector
        return get_bf_vector('gev+ban', target, noise)

    # SNR-finiteness flags of reference-channel selections that stayed on the device (one bool
    # tensor per call); `assert_finite` reads them back together -- the reference's assert
    # (beamformer.py:619), raised where the caller synchronises instead of in the middle of a step
    _pending_finite = []

    @staticmethod
    def mvdr_souden(target, noise, shard_group=None):
        """target / noise (..., F_local, D, D) -> w (..., F_local, D); the reference channel of
        every leading problem maximises the SNR summed over ALL bins (beamformer.py:601-624,
        :627-698): under bin sharding the 2 x D sums per problem are all-reduced once.  Without
        sharding nothing leaves the device: arg-max and column gather are device ops, the
        finiteness check is deferred to `assert_finite`."""
        from . import _lib, engine
        from .extraction.beamformer import _select_reference_channel_sharded
        t = _lib.torch()
        *lead, Fl, D, _ = target.shape
        eps = np.finfo(np.float64).tiny
        mat, num, den, _ = engine.mvdr_souden(
            target.to(t.complex128).reshape(-1, D, D).contiguous(),
            noise.to(t.complex128).expand(target.shape).reshape(-1, D, D).contiguous(), eps)
        if shard_group is not None:
            num, den = num.reshape(*lead, Fl, D), den.reshape(*lead, Fl, D)
            ref = _select_reference_channel_sharded(num, den, eps, shard_group)
            return select_column(mat.reshape(*lead, Fl, D, D), ref)
        # one launch: sums over the bins, arg-max, column gather (pbbss_select_reference_channel;
        # it replaced ~25 elementwise launches of the framework per call)
        L = int(np.prod(lead)) if lead else 1
        w, _, ok = engine.select_reference_channel(mat, num, den, L, Fl, eps)
        device_ops._pending_finite.append(ok)
        if len(device_ops._pending_finite) > 4096:  # a caller that never asks: keep it bounded
            device_ops.assert_finite()
        return w.reshape(*lead, Fl, D)

    @staticmethod
    def assert_finite():
        """Raise the reference's AssertionError (non-finite SNR in the reference-channel choice)
        for the `mvdr_souden` calls since the last check; one device-to-host read."""
        pend, device_ops._pending_finite = device_ops._pending_finite, []
        if pend:
            import torch
            ok = bool(torch.cat([p.reshape(-1).to(torch.bool) for p in pend]).all().item())
            assert ok, 'non-finite SNR in the automatic reference-channel selection'

    @staticmethod
    def apply_bf(w, X):
        from .extraction import apply_beamforming_vector
        return apply_beamforming_vector(w, X)


def select_column(mat, ref):
    """mat (..., F, D, D), ref int array (...): column ref[...] of every matrix -> (..., F, D)."""
    import torch
    *lead, Fl, D, _ = mat.shape
    idx = ref if torch.is_tensor(ref) else torch.as_tensor(np.asarray(ref))
    idx = idx.to(device=mat.device, dtype=torch.int64).reshape(*lead, 1, 1, 1)
    return torch.gather(mat, -1, idx.expand(*lead, Fl, D, 1)).squeeze(-1)


BEAMFORMERS = ('gev+ban', 'mvdr_souden')


def _chain_after_masks(Y, masks_fkt, mapping, ops, beamformer='gev+ban', shard_group=None):
    """Alignment -> PSD -> beamformer -> apply for utterances / bins that are local.
    Y (U, F, T, D), masks_fkt (U, F, K, T), mapping (U, K, F) for the same bins.
    shard_group: F holds one rank's block of bins ('mvdr_souden' then all-reduces the SNR sums
    of its reference-channel choice over that group; True = the default group)."""
    import torch
    kft = masks_fkt.transpose(-3, -2).contiguous()                    # (U, K, F, T)
    aligned = ops.apply_mapping(kft, mapping)                         # (U, K, F, T)
    X = Y.transpose(-2, -1).contiguous()                              # (U, F, D, T)
    psd = ops.psd(X, aligned.transpose(-3, -2).contiguous())          # (U, F, K, D, D)
    total = psd.sum(dim=-3)
    target = psd.movedim(-3, 0).contiguous()                          # (K, U, F, D, D)
    noise = (total.unsqueeze(0) - target).contiguous()
    if beamformer == 'gev+ban':
        w = ops.gev_ban(target, noise)                                # (K, U, F, D)
    elif beamformer == 'mvdr_souden':
        w = ops.mvdr_souden(target, noise, shard_group)
    else:
        raise ValueError(f'beamformer={beamformer!r}: one of {BEAMFORMERS}')
    # one launch for all classes: the observation is shared along the class axis of `w`
    # (pbbss_apply_beamforming_vector_shared); (K, U, F, T) -> a (U, K, F, T) view, no stack copy
    enhanced = ops.apply_bf(w, X).movedim(0, 1)
    return aligned, w.movedim(0, 1).contiguous(), enhanced


class graphed:
    """A launch-bound stage as ONE HIP-graph launch: `fn(*tensors) -> tensor | tuple of tensors` is run
    a few times eagerly (so that the library's workspaces have their size), captured once into a
    HIP graph (torch.cuda.CUDAGraph -- the library enqueues on torch's current stream, which is the
    capturing one), and every call afterwards copies its arguments into the captured input buffers
    and replays the graph.  For stages made of many short kernels with fixed shapes -- the
    extraction after the mixture fit (PSD -> MVDR-Souden -> apply: 12 launches, 0.126 -> 0.073 ms at
    F = 257, D = 6, K = 3 when the stage runs on its own; behind a long kernel of the same stream
    the launches are hidden anyway and a replay gains nothing, `profiles/r06_h_extraction_chain.txt`).

    * `fn` must not synchronise with the host (no status read-back: `get_gev_vector` does one, the
      device-side reference channel of `device_ops.mvdr_souden` does not) and must be shape-static.
    * The results are the SAME tensors on every call (overwritten by the next replay): copy what
      has to outlive it.
    * `device_ops.assert_finite()` sees the flags of the most recent replay.
    * If the capture fails the object falls back to calling `fn` eagerly (`self.captured` False).
    """

    def __init__(self, fn, *example, warmup=3):
        import torch
        self.fn = fn
        self.captured = False
        self.static_in = [a.clone() for a in example]
        self.graph = None
        self.out = None
        self._flags = []
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    fn(*self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            n0 = len(device_ops._pending_finite)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.out = fn(*self.static_in)
            self._flags = device_ops._pending_finite[n0:]
            self.graph = g
            self.captured = True
        except Exception as e:   # noqa: BLE001 -- anything the capture refuses: stay eager
            import warnings
            warnings.warn(f'pipeline.graphed: capture failed ({type(e).__name__}: {e}); the stage '
                          'runs eagerly', RuntimeWarning)
            self.graph = None

    def __call__(self, *args):
        if not self.captured:
            return self.fn(*args)
        for s, a in zip(self.static_in, args):
            if s.data_ptr() != a.data_ptr():
                s.copy_(a)
        self.graph.replay()
        for f in self._flags:
            if not any(f is p for p in device_ops._pending_finite):
                device_ops._pending_finite.append(f)
        return self.out


class _Laps:
    """Stage clock of `separate(stage_ms=...)`: synchronises the device after every stage (so the
    stages no longer overlap -- a diagnostic pass, not the timed one) and adds the wall time of
    the stage to the caller's dict."""

    def __init__(self, sink, device):
        import time
        self.sink, self.device, self.clock = sink, device, time.perf_counter
        self.t0 = None
        if sink is not None:
            self._sync()
            self.t0 = self.clock()

    def _sync(self):
        if getattr(self.device, 'type', 'cpu') == 'cuda':
            import torch
            torch.cuda.synchronize(self.device)

    def lap(self, name):
        if self.sink is None:
            return
        self._sync()
        now = self.clock()
        self.sink[name] = self.sink.get(name, 0.0) + (now - self.t0) * 1e3
        self.t0 = now


def separate(Y, init, iterations=100, stft_size=None, *, shard=None, group=None,
             mask_gather_dtype=None, gather_output=False, ops=device_ops, beamformer='gev+ban',
             stage_ms=None):
    """Y (U, F, T, D) complex, init (U, F, K, T): run the chain above.  A single utterance may
    come without the leading axis (Y (F, T, D), init (F, K, T)); the results then have none either.

    beamformer: 'gev+ban' (the reference's canonical recipe) or 'mvdr_souden' with the automatic
    reference channel (beamformer.py:627-698) -- the one extraction step that couples the bins:
    with shard='bins' its per-problem SNR sums are all-reduced over the group.

    shard: None (single process), 'bins', 'utterances' or 'auto' (torch.distributed initialised;
    'auto' = 'utterances' when there are at least as many utterances as ranks, else 'bins').
    stage_ms: a dict that receives this rank's wall time per stage in milliseconds (em_ms,
    gather_ms, dhtv_ms, map_gather_ms, extract_ms, out_gather_ms) with a device synchronisation
    after each -- a diagnostic pass that explains a scaling measurement, not the timed path.
    Returns dict(masks (U, K, F', T) aligned, enhanced (U, K, F', T), bf_vector (U, K, F', D),
    mapping (U, K, F)); F' = the rank's own bins for shard='bins' without gather_output, U the
    rank's own utterances for shard='utterances' without gather_output.
    """
    import torch
    if Y.ndim == 3:
        out = separate(Y[None], init[None], iterations, stft_size, shard=shard, group=group,
                       mask_gather_dtype=mask_gather_dtype, gather_output=gather_output, ops=ops,
                       beamformer=beamformer, stage_ms=stage_ms)
        return {k: v[0] for k, v in out.items()}
    U, F, T, D = Y.shape
    if stft_size is None:
        stft_size = 2 * (F - 1)
    laps = _Laps(stage_ms, getattr(Y, 'device', None))
    if shard is None:
        masks = ops.em_masks(Y, init, iterations)                     # (U, F, K, T)
        laps.lap('em_ms')
        mapping = ops.dhtv_mapping(masks.transpose(-3, -2).contiguous(), stft_size)
        laps.lap('dhtv_ms')
        aligned, w, enhanced = _chain_after_masks(Y, masks, mapping, ops, beamformer)
        if beamformer == 'mvdr_souden' and hasattr(ops, 'assert_finite'):
            ops.assert_finite()  # the reference's SNR assert, deferred to the end of the chain
        laps.lap('extract_ms')
        return dict(masks=aligned, enhanced=enhanced, bf_vector=w, mapping=mapping)

    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if shard == 'auto':
        # utterances when there is at least one per rank (no collective, and the scaling model of
        # DESIGN section 5 / profiles/r05_scaling_model.json has it ahead of the bins at every N);
        # a single utterance or a small batch can only be split along its bins
        shard = 'utterances' if U >= world else 'bins'
    if shard == 'utterances':
        assert U >= world, (U, world, 'fewer utterances than ranks: shard the bins instead')
        lo, hi = shard_bounds(U, world, rank)
        out = separate(Y[lo:hi], init[lo:hi], iterations, stft_size, ops=ops, beamformer=beamformer,
                       stage_ms=stage_ms)
        laps = _Laps(stage_ms, getattr(Y, 'device', None))
        if gather_output:
            out = {k: all_gather_bins(v.contiguous(), U, bin_axis=0, group=group)
                   for k, v in out.items()}
            laps.lap('out_gather_ms')
        return out
    assert shard == 'bins', shard

    lo, hi = shard_bounds(F, world, rank)
    Y_loc = Y[:, lo:hi].contiguous()
    if hi > lo:
        masks_loc = ops.em_masks(Y_loc, init[:, lo:hi].contiguous(), iterations)  # (U, F_loc, K, T)
    else:  # more ranks than bins
        masks_loc = torch.empty((U, 0, init.shape[-2], T), dtype=torch.float64, device=Y.device)
    laps.lap('em_ms')
    # ---- the one real exchange step of the path: masks of all bins on every rank ----------
    send = masks_loc if mask_gather_dtype is None else masks_loc.to(mask_gather_dtype)
    masks_all = all_gather_bins(send, F, bin_axis=1, group=group).to(masks_loc.dtype)
    laps.lap('gather_ms')
    # ---- DHTV needs all bins of an utterance; utterances are independent: each rank solves
    #      its share and the (U, K, F) integer mappings are all-gathered (a few KB each) --------
    ulo, uhi = shard_bounds(U, world, rank)
    K = masks_all.shape[-2]
    if uhi > ulo:
        map_loc = ops.dhtv_mapping(masks_all[ulo:uhi].transpose(-3, -2).contiguous(), stft_size)
        map_loc = map_loc.to(torch.int64).reshape(uhi - ulo, K, F)
    else:
        map_loc = torch.empty((0, K, F), dtype=torch.int64, device=Y.device)
    laps.lap('dhtv_ms')
    mapping = all_gather_bins(map_loc, U, bin_axis=0, group=group)     # (U, K, F)
    laps.lap('map_gather_ms')
    # ---- everything downstream is per bin: own bins only --------------------------------------
    sg = True if group is None else group
    if hi > lo:
        # own bins of the (own-precision) masks; with the float64 gather the mapping -- and with it
        # every output -- is bit-identical to an unsharded run (a float32 gather rounds the masks
        # the DHTV scores are computed from: near-tied scores may then pick another permutation)
        aligned, w, enhanced = _chain_after_masks(Y_loc, masks_loc, mapping[..., lo:hi].contiguous(),
                                                  ops, beamformer, sg)
    else:
        if beamformer == 'mvdr_souden':  # keep the collective schedule of the other ranks
            from .sharding import all_reduce_sum
            all_reduce_sum(torch.zeros((2, K, U, D), dtype=torch.complex128, device=Y.device),
                           None if sg is True else sg)
        aligned = torch.empty((U, K, 0, T), dtype=masks_loc.dtype, device=Y.device)
        w = torch.empty((U, K, 0, D), dtype=torch.complex128, device=Y.device)
        enhanced = torch.empty((U, K, 0, T), dtype=torch.complex128, device=Y.device)
    laps.lap('extract_ms')
    out = dict(masks=aligned, enhanced=enhanced, bf_vector=w, mapping=mapping)
    if gather_output:
        for key in ('masks', 'enhanced', 'bf_vector'):
            out[key] = all_gather_bins(out[key].contiguous(), F, bin_axis=2, group=group)
        laps.lap('out_gather_ms')
    return out
