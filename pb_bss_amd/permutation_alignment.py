"""Frequency-permutation alignment on the device (DHTV).

Mirrors pb_bss/permutation_alignment.py for the solver that follows the EM in
the canonical pipeline (`DHTVPermutationAlignment.from_stft_size(...)(mask)`,
examples/mixture_model_example.ipynb): same constructor arguments, presets,
`alignment_plan`, `calculate_mapping`, `apply_mapping`, `__call__`.
The whole plan runs in one kernel launch (csrc/dhtv.hip); masks are
(K, F, T) like the reference, or (..., K, F, T) for batches of utterances.

Device coverage: every similarity metric of the reference ('cos' -- the default of
`from_stft_size` --, 'multiply', 'euclidean') and both assignment algorithms
('greedy', 'optimal') for DHTV, `GreedyPermutationAlignment` and
`OraclePermutationAlignment` (reference :592-786), all on the device
(one wavefront per frequency bin; the greedy solver's recursion is a
permutation prefix scan), as does `_mapping_from_score_matrix`.
"""
import numpy as np

from . import _lib, engine


def interleave(*lists):
    """Round-robin over several iterables of unequal length until all are exhausted
    (reference: permutation_alignment.py:11-38): interleave([1, 2, 3], 'ab') -> 1 a 2 b 3."""
    import itertools
    gone = object()
    for group in itertools.zip_longest(*lists, fillvalue=gone):
        for item in group:
            if item is not gone:
                yield item


def sample_random_mapping(K, F, random_state=np.random):
    """A random class permutation per frequency bin -> (K, F) (reference:
    permutation_alignment.py:41-51; same draws from `random_state`: one permutation(K) per bin)."""
    return np.stack([random_state.permutation(K) for _ in range(F)], axis=1)


__all__ = ['DHTVPermutationAlignment', 'GreedyPermutationAlignment',
           'OraclePermutationAlignment', 'apply_mapping']


def _interleave(a, b):
    out = []
    for i in range(max(len(a), len(b))):
        out.extend(x[i] for x in (a, b) if i < len(x))
    return out


def _is_complex(x):
    return x.is_complex() if _lib.is_torch(x) else np.iscomplexobj(x)


def apply_mapping(mask, mapping):
    """mask (K, F, ...) [or (..., K, F, T)], reverse mapping (K, F): frequency-aligned
    mask, `mask[mapping, range(F)]` in the reference (permutation_alignment.py:54-104).

    Like the reference's fancy indexing this is a pure gather, so any dtype comes back
    unchanged: complex input (an STFT aligned with the masks' mapping) travels through the
    float64 kernel as (re, im) pairs, float32 / integer input is widened and narrowed again
    (exact for a gather)."""
    like_torch = _lib.is_torch(mask)
    t = _lib.torch()
    m = _lib.to_device(mask)
    in_dtype = m.dtype
    if m.ndim == 2:  # (K, F): trailing axis of length 1
        m = m[..., None]
    if m.is_complex():
        m = t.view_as_real(m.to(t.complex128).contiguous())
        m = m.reshape(*m.shape[:-2], m.shape[-2] * 2)  # (..., K, F, 2 T) float64
    else:
        if m.dtype == t.int64 and m.numel() and int(m.abs().max().item()) >= 2 ** 53:
            raise NotImplementedError('int64 values beyond 2**53 do not survive the float64 gather')
        m = m.to(t.float64)
    mp = _lib.to_device(mapping).to(m.device).to(t.int32)
    *lead, K, F, T = m.shape
    assert K < 20, (K, mapping.shape)
    assert tuple(mp.shape[-2:]) == (K, F), (mask.shape, mapping.shape)
    out = engine.apply_mapping(m.reshape(-1, K, F, T).contiguous(),
                               mp.expand(*lead, K, F).reshape(-1, K, F).contiguous())
    out = out.reshape(*lead, K, F, T)
    if in_dtype.is_complex:
        out = t.view_as_complex(out.reshape(*lead, K, F, T // 2, 2).contiguous())
    out = out.to(in_dtype)
    if np.ndim(mask) == 2:
        out = out[..., 0]
    return out if like_torch else _lib.to_host(out)


class _PermutationAlignment:
    def calculate_mapping(self, mask, *args, **kwargs):
        raise NotImplementedError()

    def __call__(self, mask, *args, **kwargs):
        mapping = self.calculate_mapping(mask, *args, **kwargs)
        return self.apply_mapping(mask, mapping)

    @staticmethod
    def apply_mapping(mask, mapping):
        return apply_mapping(mask, mapping)


_DEVICE_PLANS = {}  # (plan bytes, device) -> int32 (P, 3) device tensor


class DHTVPermutationAlignment(_PermutationAlignment):
    """Segment-wise centroid alignment (reference :133-355; does not solve the
    global permutation problem)."""

    def __init__(self, *, stft_size, segment_start, segment_width, segment_shift,
                 main_iterations, sub_iterations, similarity_metric='cos',
                 algorithm='greedy'):
        self.stft_size = stft_size
        self.segment_start = segment_start
        self.segment_width = segment_width
        self.segment_shift = segment_shift
        self.main_iterations = main_iterations
        self.sub_iterations = sub_iterations
        self.similarity_metric = similarity_metric
        self.algorithm = algorithm

    @classmethod
    def from_stft_size(cls, stft_size, similarity_metric='cos'):
        """Presets of the reference (:164-184)."""
        if stft_size == 512:
            start = 70
        elif stft_size == 1024:
            start = 100
        else:
            raise ValueError('There is no default for stft_size={}.', stft_size)
        return cls(stft_size=stft_size, segment_start=start, segment_width=100,
                   segment_shift=20, main_iterations=20, sub_iterations=2,
                   similarity_metric=similarity_metric)

    @property
    def alignment_plan(self):
        """[[iterations, start, end], ...] (reference :204-293): the seed segment
        with `main_iterations`, then segments grown alternately towards high
        and low frequencies with `sub_iterations` each."""
        F = self.stft_size // 2 + 1
        if self.segment_start + self.segment_width > F:
            raise ValueError(
                f'segment_start ({self.segment_start}) '
                f'+ segment_width ({self.segment_width})\n'
                f'must be smaller than stft_size // 2 + 1 ({F}),\n'
                f'but it is {self.segment_start + self.segment_width}')
        w, sh = self.segment_width, self.segment_shift
        up = [[self.sub_iterations, s, s + w]
              for s in range(self.segment_start + sh, F - w, sh)]
        down = [[self.sub_iterations, s, s + w]
                for s in range(self.segment_start - sh, 0, -sh)]
        first = [self.main_iterations, self.segment_start, self.segment_start + w]
        if up:
            up[-1][-1] = F
        else:
            first[-1] = F
        if down:
            down[-1][1] = 0
        else:
            first[1] = 0
        return [first] + _interleave(up, down)

    def _device_plan(self, F, device):
        """The alignment plan as an int32 (P, 3) device tensor: a few hundred bytes, built and
        uploaded once per (plan, device), not per call."""
        ident = (self.stft_size, self.segment_start, self.segment_width, self.segment_shift,
                 self.main_iterations, self.sub_iterations, F, str(device))
        hit = _DEVICE_PLANS.get(ident)
        if hit is not None:
            return hit
        plan = np.asarray(self.alignment_plan, dtype=np.int32)
        assert plan[:, 2].max() <= F and plan[:, 1].min() >= 0, (plan, F)
        if len(_DEVICE_PLANS) > 64:
            _DEVICE_PLANS.clear()
        _DEVICE_PLANS[ident] = _lib.to_device(plan).to(device)
        return _DEVICE_PLANS[ident]

    def calculate_mapping_async(self, mask):
        """Device-only form for callers inside a loop (the inline aligner of
        `CACGMMTrainer.fit`, reference cacgmm.py:260-267): mask float64 (U, K, F, T) contiguous
        on the device -> (reverse mapping int32 (U, K, F), status int32 (U,)), both on the
        device and NOT synchronised.  The caller reads the status words when it next has to
        wait for the device anyway and applies the rules of `calculate_mapping` then (bit
        `_lib.ST_EIG_NOCONV` alone: a team wait ran out, nothing of the launch is valid; any
        other bit: 'score matrix is infeasible')."""
        _check_metric(self.similarity_metric)
        if self.algorithm not in ('greedy', 'optimal'):
            raise ValueError(self.algorithm)
        U, K, F, T = mask.shape
        assert F % 2 == 1, (F, 'Sure? Usually F is odd.')
        assert K < 10, (K, 'Sure?')
        # masks of consecutive EM iterations mostly arrive aligned: let the library test that on
        # all segments at once before it walks the plan (pbbss_set_dhtv_probe; same results)
        before = engine.dhtv_probe(mask.device.index)  # a caller's own setting is put back
        engine.set_dhtv_probe(3, mask.device.index)  # + the aligned features are not needed
        try:
            mapping, _, st = engine.dhtv_calculate_mapping(
                mask, self._device_plan(F, mask.device), optimal=(self.algorithm == 'optimal'),
                metric=self.similarity_metric)
        finally:
            engine.set_dhtv_probe(before, mask.device.index)
        return mapping, st

    def calculate_mapping(self, mask, plot=False):
        """mask (K, F, T) [or (..., K, F, T)] -> reverse mapping (K, F) int64."""
        if plot:
            raise NotImplementedError('plot=True needs paderbox; use the reference for plots')
        _check_metric(self.similarity_metric)
        if self.algorithm not in ('greedy', 'optimal'):
            raise ValueError(self.algorithm)
        like_torch = _lib.is_torch(mask)
        t = _lib.torch()
        if _is_complex(mask):
            raise NotImplementedError(mask.dtype)  # reference :447-448 (a float64 cast would drop Im)
        m = _lib.to_device(mask, t.float64)
        *lead, K, F, T = m.shape
        assert F % 2 == 1, (F, 'Sure? Usually F is odd.')
        assert K < 10, (K, 'Sure?')
        plan_dev = self._device_plan(F, m.device)
        mu = m.reshape(-1, K, F, T).contiguous()

        def run():
            mapping, _, st = engine.dhtv_calculate_mapping(
                mu, plan_dev, optimal=(self.algorithm == 'optimal'), metric=self.similarity_metric)
            return mapping, int(np.bitwise_or.reduce(_lib.to_host(st).reshape(-1)))

        mapping, bits = run()
        if bits & _lib.ST_EIG_NOCONV and not bits & _lib.ST_NONFINITE:
            # A wait between the workgroups that share an utterance ran out: they were not on the
            # chip at the same time (other kernels of this or another process hold compute units).
            # Nothing of that launch is used; the one-workgroup kernel needs no co-residency.
            import warnings
            warnings.warn('DHTV permutation alignment: the workgroups of an utterance were not '
                          'co-resident (GPU shared with other work); running the one-workgroup '
                          'kernel instead', RuntimeWarning)
            dev = mu.device.index
            before = engine.dhtv_team(dev)
            engine.set_dhtv_team(1, dev)
            try:
                mapping, bits = run()
            finally:
                engine.set_dhtv_team(before, dev)
        if bits != 0:
            raise ValueError('score matrix is infeasible')  # reference :512-514
        mapping = mapping.reshape(*lead, K, F).to(t.int64)
        return mapping if like_torch else _lib.to_host(mapping)


_METRICS = ('cos', 'multiply', 'euclidean')  # _ScoreMatrix (:380-417)


def _check_metric(similarity_metric):
    if similarity_metric not in _METRICS:
        # the reference resolves the name with getattr(_ScoreMatrix, name) (:434-449)
        raise AttributeError(
            f"type object '_ScoreMatrix' has no attribute {similarity_metric!r}\n"
            'Suggestions: cos, euclidean, from_name, multiply')


def _mapping_from_score_matrix(score_matrix, algorithm='optimal'):
    """score_matrix (..., K, K) [reference class, mask class] -> reverse mapping (K, ...)
    (reference :469-589: 'greedy' takes the flat argmax K times, 'optimal' is the brute-force
    search in itertools.permutations order)."""
    if algorithm not in ('greedy', 'optimal'):
        raise ValueError(algorithm)
    like_torch = _lib.is_torch(score_matrix)
    t = _lib.torch()
    sc = _lib.to_device(score_matrix, t.float64)
    *F, K, K_ = sc.shape
    assert K == K_, (tuple(sc.shape), K, K_)
    mapping, st = engine.pa_mapping_from_scores(sc.reshape(-1, K, K).contiguous(),
                                                optimal=(algorithm == 'optimal'))
    if int(st.max().item()) != 0:
        raise ValueError('score matrix is infeasible')
    mapping = mapping.reshape(K, *F).to(t.int64)
    return mapping if like_torch else _lib.to_host(mapping)


class GreedyPermutationAlignment(_PermutationAlignment):
    """Aligns every frequency to its lower neighbour and chains the result
    (reference :592-701).  As in the reference, the neighbour assignment is always the
    'greedy' one (:688); `algorithm` is stored but not used by `calculate_mapping`."""

    def __init__(self, similarity_metric='euclidean', algorithm='optimal'):
        if similarity_metric not in _METRICS:
            raise ValueError(similarity_metric)  # reference :609-612
        self.similarity_metric = similarity_metric
        self.algorithm = algorithm

    def calculate_mapping(self, mask):
        """mask (K, F, T) [or (..., K, F, T)] -> reverse mapping (K, F) int64."""
        like_torch = _lib.is_torch(mask)
        t = _lib.torch()
        if _is_complex(mask):
            raise NotImplementedError(mask.dtype)  # reference _calculate_score_matrix :447-448
        m = _lib.to_device(mask, t.float64)
        *lead, K, F, T = m.shape
        assert K < 10, (K, 'Sure?')
        assert F % 2 == 1, (F, 'Sure? Usually F is odd.', tuple(m.shape))
        m = m.reshape(-1, K, F, T).contiguous()
        mapping = t.empty((m.shape[0], K, F), dtype=t.int32, device=m.device)
        if F > 1:
            _, _, st = engine.pa_pairwise_mapping(m[:, :, 1:], m[:, :, :-1],
                                                  self.similarity_metric, optimal=False,
                                                  mapping=mapping, col0=1)
            if int(st.max().item()) != 0:
                raise ValueError('score matrix is infeasible')
        engine.pa_compose_mapping(mapping)
        mapping = mapping.reshape(*lead, K, F).to(t.int64)
        return mapping if like_torch else _lib.to_host(mapping)


class OraclePermutationAlignment(_PermutationAlignment):
    """Aligns every frequency of `mask` to the same frequency of `reference_mask`
    (reference :703-786)."""

    def __init__(self, similarity_metric='euclidean', algorithm='optimal'):
        assert algorithm in ['greedy', 'optimal'], algorithm
        _check_metric(similarity_metric)
        self.similarity_metric = similarity_metric
        self.algorithm = algorithm

    def calculate_mapping(self, mask, reference_mask):
        """mask, reference_mask (K, F, T) or (K, T) -> reverse mapping (K, F) / (K,) int64."""
        like_torch = _lib.is_torch(mask)
        t = _lib.torch()
        if _is_complex(mask) or _is_complex(reference_mask):
            raise NotImplementedError(mask.dtype, reference_mask.dtype)  # reference :447-448
        m = _lib.to_device(mask, t.float64)
        r = _lib.to_device(reference_mask, t.float64).to(m.device)
        assert tuple(m.shape) == tuple(r.shape), (tuple(m.shape), tuple(r.shape))
        K, *F, T = m.shape
        assert K < 10, (K, 'Sure?')
        if len(F) == 1:
            assert F[0] % 2 == 1, (F, 'Sure? Usually F is odd.', tuple(m.shape))
        if len(F) > 1:
            raise NotImplementedError(
                'more than one independent axis: reshape to (K, F, T) first')
        nF = F[0] if F else 1
        mapping, _, st = engine.pa_pairwise_mapping(
            m.reshape(1, K, nF, T).contiguous(), r.reshape(1, K, nF, T).contiguous(),
            self.similarity_metric, optimal=(self.algorithm == 'optimal'))
        if int(st.max().item()) != 0:
            raise ValueError('score matrix is infeasible')
        mapping = mapping.reshape(K, *F).to(t.int64)
        return mapping if like_torch else _lib.to_host(mapping)
