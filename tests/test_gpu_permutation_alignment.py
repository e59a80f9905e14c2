"""GPU: DHTV permutation alignment kernel against vectors of
the real reference (tests/golden/dhtv_alignment.npz) and the NumPy oracle.
Integer result: the mapping must be identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_dhtv_mapping_identical_to_reference():
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    g = np.load(os.path.join(GOLDEN, 'dhtv_alignment.npz'))
    for tag, size in [('s512_k2', 512), ('s1024_k3', 1024)]:
        solver = DHTVPermutationAlignment.from_stft_size(size)
        assert (np.asarray(solver.alignment_plan) == g[tag + '_plan']).all()
        mask = g[tag + '_mask'].astype(np.float64)
        mapping = solver.calculate_mapping(mask)
        assert mapping.dtype == np.int64 and mapping.shape == g[tag + '_mapping'].shape
        assert (mapping == g[tag + '_mapping']).all()
        aligned = solver(mask)
        assert np.allclose(aligned.sum(-1), g[tag + '_aligned_sum'])
        assert (aligned == mask[mapping, range(mask.shape[1])]).all()
        solver.algorithm = 'optimal'
        assert (solver.calculate_mapping(mask) == g[tag + '_mapping_optimal']).all()
        for metric in ('multiply', 'euclidean'):
            for alg in ('greedy', 'optimal'):
                sv = DHTVPermutationAlignment.from_stft_size(size, metric)
                sv.algorithm = alg
                assert (sv.calculate_mapping(mask) == g[f'{tag}_mapping_{metric}_{alg}']).all()


def test_dhtv_batch_and_plan_doctests():
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from pb_bss_amd import engine
    from oracle import permutation_alignment as op
    assert DHTVPermutationAlignment.from_stft_size(512).alignment_plan == [
        [20, 70, 170], [2, 90, 190], [2, 50, 150], [2, 110, 210], [2, 30, 130],
        [2, 130, 230], [2, 0, 110], [2, 150, 257]]  # permutation_alignment.py:218-227
    assert DHTVPermutationAlignment(stft_size=512, segment_start=0, segment_width=257,
                                    segment_shift=20, main_iterations=20,
                                    sub_iterations=2).alignment_plan == [[20, 0, 257]]
    with pytest.raises(ValueError):
        DHTVPermutationAlignment(stft_size=512, segment_start=70, segment_width=300,
                                 segment_shift=20, main_iterations=20,
                                 sub_iterations=2).alignment_plan
    with pytest.raises(ValueError):
        DHTVPermutationAlignment.from_stft_size(256)
    rng = np.random.default_rng(3)
    U, K, F, T = 3, 3, 257, 64
    masks = rng.uniform(size=(U, K, F, T)) ** 3
    solver = DHTVPermutationAlignment.from_stft_size(512)
    mapping = solver.calculate_mapping(masks)
    plan = op.alignment_plan(512, **op.PRESETS[512])
    for u in range(U):
        assert (mapping[u] == op.dhtv_calculate_mapping(masks[u], plan)).all()
    # every metric of the reference, on noisy masks where the metrics disagree
    noisy = rng.uniform(size=(2, K, F, T)) ** 2
    seen = []
    for metric in ('cos', 'multiply', 'euclidean'):
        for team in (1, 0):
            engine.set_dhtv_team(team)
            mp = DHTVPermutationAlignment.from_stft_size(512, metric).calculate_mapping(noisy)
            for u in range(2):
                assert (mp[u] == op.dhtv_calculate_mapping(noisy[u], plan, 'greedy', metric)).all(), \
                    (metric, team, u)
        seen.append(mp)
    engine.set_dhtv_team(0)
    assert any((seen[0] != x).any() for x in seen[1:])
    with pytest.raises(AttributeError):
        DHTVPermutationAlignment.from_stft_size(512, 'coss').calculate_mapping(masks[0])


def test_em_masks_align_end_to_end():
    """fit -> predict -> device DHTV on the masks == oracle DHTV on the same masks."""
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from oracle import permutation_alignment as op, synth
    Y, init = synth.make_stft(257, 120, 4, 2, seed=5)
    masks = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=10)
    kft = np.ascontiguousarray(masks.transpose(1, 0, 2))
    solver = DHTVPermutationAlignment.from_stft_size(512)
    mapping = solver.calculate_mapping(kft)
    assert (mapping == op.dhtv_calculate_mapping(kft, solver.alignment_plan)).all()


@pytest.mark.parametrize('K,F,T,stft', [(2, 257, 130, 512), (3, 513, 500, 1024), (4, 257, 300, 512)])
def test_team_kernel_equals_single_workgroup_kernel_and_oracle(K, F, T, stft):
    """few utterances: several workgroups share one utterance (pbbss_set_dhtv_team);
    every team size must reproduce the one-workgroup kernel's mapping and the oracle's"""
    from pb_bss_amd import engine
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from oracle import permutation_alignment as op
    rng = np.random.default_rng(K * 100 + F)
    act = rng.uniform(size=(2, K, 1, T)) ** 4
    mask = act * rng.uniform(0.5, 1.0, size=(2, K, F, T)) + 0.05 * rng.uniform(size=(2, K, F, T))
    mask /= mask.sum(1, keepdims=True)
    for u in range(2):
        for f in range(F):
            mask[u, :, f] = mask[u, rng.permutation(K), f]
    solver = DHTVPermutationAlignment.from_stft_size(stft)
    want = np.stack([op.dhtv_calculate_mapping(mask[u], solver.alignment_plan) for u in range(2)])
    try:
        # 1: one workgroup; >= 2: frame-slice kernel; <= -2: bin-chunk team kernel
        for team in (1, 2, 5, 16, 32, 0, -2, -5, -16, -32):
            engine.set_dhtv_team(team)
            got = solver.calculate_mapping(mask)
            assert np.array_equal(got, want), team
            assert np.array_equal(solver.calculate_mapping(mask[0]), want[0]), team
    finally:
        engine.set_dhtv_team(0)


@pytest.mark.parametrize('K,F,T,plan', [
    (3, 257, 200, [[20, 0, 257]]),                      # full-width window: streamed passes
    (5, 129, 333, [[6, 10, 90], [2, 0, 60], [2, 50, 129]]),  # 25 scores per bin, ragged frames
    (2, 65, 1000, [[4, 0, 65], [3, 5, 6]]),             # one-bin segment, 16 slices
    (4, 257, 128, [[20, 70, 170], [2, 0, 110], [2, 150, 257]]),
    (1, 65, 200, [[3, 0, 65]]),                          # one class: nothing to permute
    (3, 33, 65, [[5, 0, 33], [2, 3, 20]]),              # second workgroup holds ONE frame
])
def test_frame_slice_kernel_shapes_metrics_and_features(K, F, T, plan):
    """frame-slice kernel (pbbss_set_dhtv_team >= 2): every metric and both solvers against
    the oracle, the aligned unit-norm features it leaves in the scratch, several team sizes"""
    from pb_bss_amd import _lib, engine
    from oracle import permutation_alignment as op
    rng = np.random.default_rng(K * 1000 + T)
    U = 2
    act = rng.uniform(size=(U, K, 1, T)) ** 4
    mask = act * rng.uniform(0.3, 1.0, size=(U, K, F, T)) + 0.1 * rng.uniform(size=(U, K, F, T))
    for u in range(U):
        for f in range(F):
            mask[u, :, f] = mask[u, rng.permutation(K), f]
    md = _lib.to_device(mask)
    pd = _lib.to_device(np.asarray(plan, np.int32))
    try:
        for team in (2, 4, 64):
            engine.set_dhtv_team(team)
            for metric in ('cos', 'multiply', 'euclidean'):
                for optimal in (False, True):
                    mapping, feat, st = engine.dhtv_calculate_mapping(md, pd, optimal, metric)
                    assert (_lib.to_host(st) == 0).all()
                    got, fh = _lib.to_host(mapping), _lib.to_host(feat)
                    for u in range(U):
                        want = op.dhtv_calculate_mapping(
                            mask[u], plan, 'optimal' if optimal else 'greedy', metric)
                        assert np.array_equal(got[u], want), (team, metric, optimal, u)
                        al = mask[u][want, range(F)]
                        if metric == 'cos':
                            al = al / np.maximum(np.linalg.norm(al, axis=-1, keepdims=True),
                                                 np.finfo(np.float64).tiny)
                        assert np.abs(fh[u] - al).max() < 1e-14 * max(1.0, np.abs(al).max()), \
                            (team, metric)
        # a non-finite mask entry is reported like the reference's 'score matrix is infeasible'
        bad = mask.copy()
        bad[1, 0, plan[0][1], 3] = np.nan
        engine.set_dhtv_team(4)
        _, _, st = engine.dhtv_calculate_mapping(_lib.to_device(bad), pd)
        st = _lib.to_host(st)
        assert st[0] == 0 and st[1] != 0
    finally:
        engine.set_dhtv_team(0)


def test_pairwise_solvers_identical_to_reference():
    """GreedyPermutationAlignment / OraclePermutationAlignment / _mapping_from_score_matrix on
    the device against vectors of the real reference: integer mappings must be identical."""
    from pb_bss_amd.permutation_alignment import (GreedyPermutationAlignment,
                                                  OraclePermutationAlignment,
                                                  _mapping_from_score_matrix)
    from pb_bss_amd import _lib, engine
    g = np.load(os.path.join(GOLDEN, 'pairwise_alignment.npz'))
    for tag in ('k2', 'k3', 'k4'):
        m = g[tag + '_mask'].astype(np.float64)
        r = g[tag + '_reference'].astype(np.float64)
        for metric in ('cos', 'multiply', 'euclidean'):
            _, sc, _ = engine.pa_pairwise_mapping(
                _lib.to_device(m)[None], _lib.to_device(r)[None], metric, False, want_scores=True)
            ref_sc = g[f'{tag}_{metric}_scores']
            assert np.abs(_lib.to_host(sc)[0] - ref_sc).max() < 1e-12 * max(1.0, np.abs(ref_sc).max())
            mp = GreedyPermutationAlignment(similarity_metric=metric).calculate_mapping(m)
            assert mp.dtype == np.int64 and (mp == g[f'{tag}_{metric}_greedy']).all(), (tag, metric)
            for alg in ('greedy', 'optimal'):
                solver = OraclePermutationAlignment(similarity_metric=metric, algorithm=alg)
                mo = solver.calculate_mapping(m, r)
                assert (mo == g[f'{tag}_{metric}_oracle_{alg}']).all(), (tag, metric, alg)
                assert (solver(m, r) == m[mo, range(m.shape[1])]).all()
    m, r = g['blocks_mask'], g['blocks_reference']
    assert (GreedyPermutationAlignment().calculate_mapping(m) == g['blocks_greedy']).all()
    assert (OraclePermutationAlignment().calculate_mapping(m, r) == g['blocks_oracle']).all()
    assert (GreedyPermutationAlignment('cos').calculate_mapping(m) == g['blocks_greedy_cos']).all()
    assert (OraclePermutationAlignment('cos', 'greedy').calculate_mapping(m, r)
            == g['blocks_oracle_cos_greedy']).all()
    assert (OraclePermutationAlignment()(m, r) == r).all()
    for alg in ('greedy', 'optimal'):
        assert (_mapping_from_score_matrix(g['scores'], alg) == g['scores_' + alg]).all()
    sm = np.array([[11, 10, 0], [4, 5, 10], [6, 0, 5]])  # doctest, permutation_alignment.py:475-508
    assert (_mapping_from_score_matrix(sm, 'optimal') == [1, 2, 0]).all()
    assert (_mapping_from_score_matrix(sm, 'greedy') == [0, 2, 1]).all()
    with pytest.raises(ValueError, match='infeasible'):
        _mapping_from_score_matrix(np.array([[1.0, np.nan], [0.0, 1.0]]), 'greedy')
    with pytest.raises(ValueError, match='infeasible'):
        OraclePermutationAlignment().calculate_mapping(np.full((2, 3, 4), np.inf), np.zeros((2, 3, 4)))
    with pytest.raises(ValueError):
        GreedyPermutationAlignment(similarity_metric='coss')
    with pytest.raises(AttributeError):
        OraclePermutationAlignment(similarity_metric='coss')


@pytest.mark.parametrize('K,F,T', [(2, 257, 130), (3, 513, 500), (5, 129, 77), (8, 65, 40)])
def test_pairwise_solvers_match_oracle_at_size(K, F, T):
    """Config-2-sized masks (and K up to 8, batches, the (K, T) form) against the NumPy oracle;
    the greedy solver's permutation scan against the sequential recursion."""
    from pb_bss_amd.permutation_alignment import (GreedyPermutationAlignment,
                                                  OraclePermutationAlignment)
    from oracle import permutation_alignment as op
    rng = np.random.default_rng(K * 1000 + F)
    ref = rng.uniform(size=(K, 1, T)) ** 4 * rng.uniform(0.5, 1.0, size=(K, F, T)) \
        + 0.05 * rng.uniform(size=(K, F, T))
    ref /= ref.sum(0, keepdims=True)
    perm = np.stack([rng.permutation(K) for _ in range(F)], 1)
    mask = ref[perm, range(F)] * rng.uniform(0.9, 1.1, size=(K, F, T))
    for metric in ('cos', 'multiply', 'euclidean'):
        mg = GreedyPermutationAlignment(similarity_metric=metric).calculate_mapping(mask)
        assert (mg == op.greedy_calculate_mapping(mask, metric)).all(), metric
        # every column is a permutation
        assert (np.sort(mg, axis=0) == np.arange(K)[:, None]).all()
        algs = ('greedy', 'optimal') if K <= 5 else ('greedy',)
        for alg in algs:
            mo = OraclePermutationAlignment(metric, alg).calculate_mapping(mask, ref)
            assert (mo == op.oracle_calculate_mapping(mask, ref, metric, alg)).all(), (metric, alg)
    # the oracle solver recovers the permutation that was applied (noise is mild)
    mo = OraclePermutationAlignment('cos').calculate_mapping(mask, ref) if K <= 5 else \
        OraclePermutationAlignment('cos', 'greedy').calculate_mapping(mask, ref)
    assert (np.argsort(perm, axis=0) == mo).all()
    # batch of utterances on the greedy solver, and the (K, T) form of the oracle solver
    batch = np.stack([mask, ref, mask[::-1]])
    mb = GreedyPermutationAlignment('cos').calculate_mapping(batch)
    for u in range(3):
        assert (mb[u] == op.greedy_calculate_mapping(batch[u], 'cos')).all()
    m1 = OraclePermutationAlignment('euclidean', 'greedy').calculate_mapping(mask[:, 0], ref[:, 0])
    assert m1.shape == (K,)
    assert (m1 == op.oracle_calculate_mapping(mask[:, :1], ref[:, :1], 'euclidean', 'greedy')[:, 0]).all()


def test_team_timeout_falls_back_to_the_one_workgroup_kernel(monkeypatch):
    """A timed-out wait between the workgroups of an utterance (status bit EIG_NOCONV: they were
    not co-resident) is not an infeasible score matrix: the aligner warns, reruns with the
    one-workgroup kernel and restores the team setting."""
    import torch
    from oracle import permutation_alignment as op
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    rng = np.random.default_rng(11)
    K, F, T = 3, 257, 300
    act = rng.uniform(size=(K, T)) ** 4
    mask = act[:, None, :] * rng.uniform(0.5, 1.0, size=(K, F, T)) + 0.05 * rng.uniform(size=(K, F, T))
    mask /= mask.sum(0, keepdims=True)
    perm = np.stack([rng.permutation(K) for _ in range(F)], 1)
    mask = mask[perm, np.arange(F)]
    want = op.dhtv_calculate_mapping(mask, op.alignment_plan(512, **op.PRESETS[512]))

    real = engine.dhtv_calculate_mapping
    teams, calls = [], []
    real_set = engine.set_dhtv_team

    def fake_set(v, device_index=None):
        teams.append(v)
        return real_set(v, device_index)

    def fake(*a, **kw):
        mapping, feat, st = real(*a, **kw)
        calls.append(engine.dhtv_team(a[0].device.index))
        if len(calls) == 1:
            st = st | _lib.ST_EIG_NOCONV   # as a timed-out team launch reports it
            mapping = torch.zeros_like(mapping)
        return mapping, feat, st

    monkeypatch.setattr(engine, 'dhtv_calculate_mapping', fake)
    monkeypatch.setattr(engine, 'set_dhtv_team', fake_set)
    engine.set_dhtv_team(0)
    with pytest.warns(RuntimeWarning, match='co-resident'):
        got = DHTVPermutationAlignment.from_stft_size(512).calculate_mapping(_lib.to_device(mask))
    assert (_lib.to_host(got) == want).all()
    assert calls == [0, 1] and teams == [0, 1, 0]
    # a non-finite mask stays the reference's error
    bad = mask.copy()
    bad[0, 3, 5] = np.nan
    with pytest.raises(ValueError, match='infeasible'):
        DHTVPermutationAlignment.from_stft_size(512).calculate_mapping(_lib.to_device(bad))


def _structured_masks(K, F, T, seed, permuted_bins=()):
    """masks with one activity pattern per class (aligned), the class rows of `permuted_bins`
    rotated by one"""
    rng = np.random.default_rng(seed)
    act = rng.uniform(size=(K, 1, T)) ** 4
    mask = act * rng.uniform(0.5, 1.0, size=(K, F, T)) + 0.05 * rng.uniform(size=(K, F, T))
    mask /= mask.sum(0, keepdims=True)
    for f in permuted_bins:
        mask[:, f] = np.roll(mask[:, f], 1, axis=0)
    return mask


@pytest.mark.parametrize('K,F,T,stft', [(3, 513, 500, 1024), (2, 257, 300, 512), (4, 257, 200, 512)])
@pytest.mark.parametrize('metric,algorithm', [('cos', 'greedy'), ('euclidean', 'optimal')])
def test_identity_probe_changes_nothing_but_the_time(K, F, T, stft, metric, algorithm):
    """pbbss_set_dhtv_probe: all segments evaluated at once in front of the plan.  Aligned masks
    -> identity without walking the plan; one flipped bin anywhere (first segment, an outer
    segment, the last bin) or fully shuffled masks -> the plan runs; always the mapping of the
    plain kernel and of the oracle."""
    import torch
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from oracle import permutation_alignment as op
    solver = DHTVPermutationAlignment.from_stft_size(stft, metric)
    solver.algorithm = algorithm
    plan = op.alignment_plan(stft, **op.PRESETS[stft])
    rng = np.random.default_rng(5)
    cases = {
        'aligned': _structured_masks(K, F, T, 40),
        'first_segment_bin': _structured_masks(K, F, T, 41, [plan[0][1] + 3]),
        'outer_bins': _structured_masks(K, F, T, 42, [1, F - 1]),
        'shuffled': _structured_masks(K, F, T, 43, rng.choice(F, F // 2, replace=False)),
    }
    for tag, mask in cases.items():
        want = op.dhtv_calculate_mapping(mask, plan, algorithm=algorithm, similarity_metric=metric)
        plain = solver.calculate_mapping(mask)
        m = _lib.to_device(mask, torch.float64)[None].contiguous()
        probed, st = solver.calculate_mapping_async(m)
        probed = _lib.to_host(probed)[0]
        assert int(_lib.to_host(st)[0]) == 0, tag
        assert np.array_equal(plain, want), tag
        assert np.array_equal(probed, want), tag
        if tag == 'aligned':
            assert np.array_equal(want, np.repeat(np.arange(K)[:, None], F, axis=1))
        else:
            assert not np.array_equal(want, np.repeat(np.arange(K)[:, None], F, axis=1)), tag
    # non-finite input is reported with the probe in front as well
    bad = cases['aligned'].copy()
    bad[0, 5, 7] = np.nan
    _, st = solver.calculate_mapping_async(_lib.to_device(bad, torch.float64)[None].contiguous())
    assert int(_lib.to_host(st)[0]) & _lib.ST_NONFINITE


def test_identity_probe_with_two_utterances_and_sparse_plans():
    """Probe flags are per (utterance, segment): one aligned and one shuffled utterance in the same
    call; plus a hand-made plan with a zero-iteration segment and overlapping windows."""
    import torch
    from pb_bss_amd import _lib
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from oracle import permutation_alignment as op
    K, F, T = 3, 257, 300
    rng = np.random.default_rng(7)
    masks = np.stack([_structured_masks(K, F, T, 50),
                      _structured_masks(K, F, T, 51, rng.choice(F, 40, replace=False))])
    for solver in (DHTVPermutationAlignment.from_stft_size(512),
                   DHTVPermutationAlignment(stft_size=512, segment_start=40, segment_width=60,
                                            segment_shift=30, main_iterations=3, sub_iterations=1)):
        plan = solver.alignment_plan
        got, st = solver.calculate_mapping_async(_lib.to_device(masks, torch.float64).contiguous())
        got = _lib.to_host(got)
        assert not _lib.to_host(st).any()
        for u in range(2):
            assert np.array_equal(got[u], op.dhtv_calculate_mapping(masks[u], plan)), u
        assert np.array_equal(got[0], np.repeat(np.arange(K)[:, None], F, axis=1))
        assert not np.array_equal(got[1], got[0])
    # a raw plan with a segment of zero iterations (skipped by the reference's range(0) loop)
    from pb_bss_amd import engine
    plan = np.array([[4, 60, 160], [0, 0, 120], [2, 100, 257], [2, 0, 100]], dtype=np.int32)
    m = _lib.to_device(masks, torch.float64).contiguous()
    engine.set_dhtv_probe(True)
    try:
        got, _, st = engine.dhtv_calculate_mapping(m, _lib.to_device(plan))
    finally:
        engine.set_dhtv_probe(False)
    want, _, _ = engine.dhtv_calculate_mapping(m, _lib.to_device(plan))
    assert not _lib.to_host(st).any()
    assert np.array_equal(_lib.to_host(got), _lib.to_host(want))
    for u in range(2):
        assert np.array_equal(_lib.to_host(got)[u], op.dhtv_calculate_mapping(masks[u], plan.tolist()))
