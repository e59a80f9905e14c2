"""NumPy restatement of the reference DHTV permutation alignment
(TEST INFRASTRUCTURE).  Citations: /root/reference/pb_bss/permutation_alignment.py.
Pinned by oracle/make_golden.py -> tests/golden/dhtv_*.npz.
"""
import itertools

import numpy as np

__all__ = ['alignment_plan', 'mapping_from_score_matrix', 'dhtv_calculate_mapping',
           'apply_mapping', 'PRESETS', 'score_matrix', 'mapping_from_score_matrices',
           'greedy_calculate_mapping', 'oracle_calculate_mapping']

# from_stft_size presets, permutation_alignment.py:164-184
PRESETS = {
    512: dict(segment_start=70, segment_width=100, segment_shift=20,
              main_iterations=20, sub_iterations=2),
    1024: dict(segment_start=100, segment_width=100, segment_shift=20,
               main_iterations=20, sub_iterations=2),
}


def _interleave(a, b):
    out = []
    for i in range(max(len(a), len(b))):
        if i < len(a):
            out.append(a[i])
        if i < len(b):
            out.append(b[i])
    return out


def alignment_plan(stft_size, segment_start, segment_width, segment_shift,
                   main_iterations, sub_iterations):
    """[(iterations, start, end), ...]: the seed segment first, then segments
    growing alternately upwards and downwards (permutation_alignment.py:204-293)."""
    F = stft_size // 2 + 1
    if segment_start + segment_width > F:
        raise ValueError(
            f'segment_start ({segment_start}) + segment_width ({segment_width})\n'
            f'must be smaller than stft_size // 2 + 1 ({F}),\n'
            f'but it is {segment_start + segment_width}')
    up = [[sub_iterations, s, s + segment_width]
          for s in range(segment_start + segment_shift, F - segment_width, segment_shift)]
    down = [[sub_iterations, s, s + segment_width]
            for s in range(segment_start - segment_shift, 0, -segment_shift)]
    first = [main_iterations, segment_start, segment_start + segment_width]
    if up:
        up[-1][-1] = F
    else:
        first[-1] = F
    if down:
        down[-1][1] = 0
    else:
        first[1] = 0
    return [first] + _interleave(up, down)


def _unit(a):
    """permutation_alignment.py:358-377: a / max(||a||, tiny) along the last axis."""
    n = np.linalg.norm(a, axis=-1, keepdims=True)
    return a / np.maximum(n, np.finfo(n.dtype).tiny)


def mapping_from_score_matrix(score, algorithm='greedy'):
    """permutation_alignment.py:469-589 for one (K, K) score matrix
    (rows = reference/centroid class, columns = mask class)."""
    score = np.asarray(score, dtype=np.float64)
    if not np.all(np.isfinite(score)):
        raise ValueError('score matrix is infeasible')
    K = score.shape[0]
    if algorithm == 'greedy':
        s = score.copy()
        out = np.zeros(K, dtype=int)
        for _ in range(K):
            i, j = np.unravel_index(np.argmax(s), s.shape)
            s[i, :] = -np.inf
            s[:, j] = -np.inf
            out[i] = j
        return out
    if algorithm == 'optimal':
        best, best_p = -np.inf, None
        for p in itertools.permutations(range(K)):
            v = sum(score[range(K), p])
            if v > best:
                best, best_p = v, p
        return np.array(best_p)
    raise ValueError(algorithm)


def dhtv_calculate_mapping(mask, plan, algorithm='greedy', similarity_metric='cos'):
    """permutation_alignment.py:295-355.  mask (K, F, T) -> reverse mapping (K, F).
    'cos': unit-norm features and centroid, scored with `multiply` (:155-160, :309-312,
    :337-341); 'multiply' / 'euclidean': raw features and centroid."""
    K, F, _ = mask.shape
    assert F % 2 == 1, (F, 'Sure? Usually F is odd.')
    cos = similarity_metric == 'cos'
    feat = np.array(mask, dtype=np.float64)
    if cos:
        feat = _unit(feat)
    mapping = np.repeat(np.arange(K)[:, None], F, axis=1)
    ident = np.arange(K)
    for iterations, start, end in plan:
        for _ in range(iterations):
            cent = np.mean(feat[:, start:end, :], axis=1)
            if cos:
                cent = _unit(cent)
            changed = False
            for f in range(start, end):
                if similarity_metric == 'euclidean':
                    score = -np.sqrt(np.sum(
                        (feat[:, None, f, :] - cent[None]) ** 2, axis=-1)).T
                elif similarity_metric in ('cos', 'multiply'):
                    score = np.einsum('KT,kT->kK', feat[:, f, :], cent)
                else:
                    raise AttributeError(similarity_metric)
                perm = mapping_from_score_matrix(score, algorithm)
                if not (perm == ident).all():
                    changed = True
                    feat[:, f, :] = feat[perm, f, :]
                    mapping[:, f] = mapping[perm, f]
            if not changed:
                break
    return mapping


def apply_mapping(mask, mapping):
    """permutation_alignment.py:54-104: mask (K, F, ...), mapping (K, F)."""
    K, F = mapping.shape
    return mask[mapping, range(F)]


def score_matrix(mask, reference_mask, similarity_metric):
    """permutation_alignment.py:396-417 (_ScoreMatrix.cos / multiply / euclidean) for
    mask, reference_mask (K, F, T) -> (F, K, K) indexed [bin, reference class, mask class]."""
    mask = np.asarray(mask, dtype=np.float64)
    reference_mask = np.asarray(reference_mask, dtype=np.float64)
    if similarity_metric == 'cos':
        mask, reference_mask = _unit(mask), _unit(reference_mask)
    if similarity_metric in ('cos', 'multiply'):
        return np.einsum('KFT,kFT->FkK', mask, reference_mask)
    if similarity_metric == 'euclidean':
        # -sqrt(sum |mask[:, None] - ref[None]|^2, -1).T : (K_mask, k_ref, F) reversed
        d = np.sqrt(np.sum((mask[:, None] - reference_mask[None]) ** 2, axis=-1))
        return -np.transpose(d, (2, 1, 0))
    raise AttributeError(similarity_metric)


def mapping_from_score_matrices(scores, algorithm):
    """permutation_alignment.py:469-589 over a stack (F, K, K) -> (K, F)."""
    scores = np.asarray(scores)
    if not np.all(np.isfinite(scores)):
        raise ValueError('score matrix is infeasible')
    return np.stack([mapping_from_score_matrix(s, algorithm) for s in scores], axis=1)


def greedy_calculate_mapping(mask, similarity_metric='euclidean'):
    """GreedyPermutationAlignment.calculate_mapping, permutation_alignment.py:614-701:
    every bin against its lower neighbour with the 'greedy' assignment (:687-688), identity
    for bin 0 (:690-691), then mapping[:, f] = mapping[mapping[:, f-1], f] (:694-695)."""
    K, F, T = mask.shape
    assert K < 10 and F % 2 == 1
    scores = score_matrix(mask[:, 1:, :], mask[:, :-1, :], similarity_metric)
    mapping = mapping_from_score_matrices(scores, 'greedy') if F > 1 else np.zeros((K, 0), int)
    mapping = np.append(np.arange(K, dtype=mapping.dtype)[:, None], mapping, axis=-1)
    for f in range(1, F):
        mapping[:, f] = mapping[mapping[:, f - 1], f]
    return mapping


def oracle_calculate_mapping(mask, reference_mask, similarity_metric='euclidean',
                             algorithm='optimal'):
    """OraclePermutationAlignment.calculate_mapping, permutation_alignment.py:711-786."""
    assert mask.shape == reference_mask.shape
    return mapping_from_score_matrices(score_matrix(mask, reference_mask, similarity_metric),
                                       algorithm)
