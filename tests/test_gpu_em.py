"""GPU parity of the persistent cACGMM EM kernel against the NumPy oracle
(oracle/cacgmm.py, itself pinned to the reference by tests/golden).

Protocol: the device gets the complex64 observation and a
float64 initialisation; the oracle gets the exact complex128 upcast of the same
values, so both sides compute in float64 on identical inputs.
Tolerances: single step 1e-9, trajectories 1e-5 (BASELINE.json north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    from pb_bss_amd import _lib
    return _lib.to_device(x, dtype)


def _host(x):
    from pb_bss_amd import _lib
    return _lib.to_host(x)


def _oracle_fit(Y, init, iterations, **kw):
    from oracle import cacgmm as oc
    Y128 = Y.astype(np.complex128)
    m = oc.em_fit(Y128, init, iterations=iterations, **kw)
    return m, oc.em_predict(m, Y128)


def _device_fit(Y, init, iterations, **kw):
    from pb_bss_amd import engine
    K = init.shape[-2]
    return engine.em_fit(_dev(Y), K, gamma0=_dev(init), iterations=iterations,
                         final_predict=True, return_q=True, **kw)


def _cov(vec, val):
    return np.einsum('...wx,...x,...zx->...wz', vec, val, vec.conj())


@pytest.mark.parametrize('F,T,D,K', [(5, 64, 2, 2), (9, 200, 3, 2), (17, 130, 6, 3),
                                     (33, 500, 8, 3), (4, 257, 8, 4), (3, 100, 7, 1)])
def test_single_m_step_and_predict(F, T, D, K):
    """iterations=1: weighted covariance + eigh + floor, then one E-step."""
    from oracle import synth
    Y, init = synth.make_stft(F, T, D, max(K, 2), seed=F)
    init = init[:, :K] / init[:, :K].sum(axis=1, keepdims=True)
    m, mask = _oracle_fit(Y, init, 1)
    r = _device_fit(Y, init, 1)
    assert np.abs(_host(r['weight'])[..., None] - m['weight']).max() < 1e-14
    cov = _cov(_host(r['eigvec']), _host(r['eigval']))
    ref = _cov(m['eigvec'], m['eigval'])
    assert np.abs(cov - ref).max() < 1e-12
    assert np.abs(_host(r['eigval']) - m['eigval']).max() < 1e-12
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-9


@pytest.mark.parametrize('iters', [2, 5, 20])
def test_trajectory_small(iters):
    from oracle import synth
    Y, init = synth.make_stft(65, 300, 8, 3, seed=0)
    m, mask = _oracle_fit(Y, init, iters)
    r = _device_fit(Y, init, iters)
    assert (_host(r['status']) & 3 == 0).all()
    err = np.abs(_host(r['affiliation']) - mask).max()
    assert err < 1e-7, err
    assert np.abs(_host(r['weight'])[..., None] - m['weight']).max() < 1e-8


def test_config1_plumbing_shape():
    """BASELINE config 1: F=129, T=200, D=2, K=2, 20 iterations."""
    from oracle import synth
    Y, init = synth.make_stft(129, 200, 2, 2, seed=0)
    m, mask = _oracle_fit(Y, init, 20)
    r = _device_fit(Y, init, 20)
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-8


def test_config2_full_100_iterations():
    """BASELINE config 2 (the headline): F=513, T=500, D=8, K=3, 100 iterations.
    The contract (BASELINE north_star) is 1e-5 max-abs against the float64 reference path; the
    measured error is ~2e-11, so the assertion sits at 1e-8 -- three decades of regression
    head-room are kept, not six."""
    from oracle import synth
    CONTRACT_TOL = 1e-5
    Y, init = synth.make_stft(513, 500, 8, 3, seed=0)
    m, mask = _oracle_fit(Y, init, 100)
    r = _device_fit(Y, init, 100)
    err = np.abs(_host(r['affiliation']) - mask).max()
    assert err < 1e-8 < CONTRACT_TOL, err
    cov = _cov(_host(r['eigvec']), _host(r['eigval']))
    assert np.abs(cov - _cov(m['eigvec'], m['eigval'])).max() < 1e-8


def test_force_eig_equals_fast_path():
    """The Cholesky fast path and the per-iteration Jacobi path are the same
    algorithm up to rounding."""
    from oracle import synth
    Y, init = synth.make_stft(33, 250, 8, 3, seed=3)
    a = _device_fit(Y, init, 10)
    b = _device_fit(Y, init, 10, force_eig=True)
    assert np.abs(_host(a['affiliation']) - _host(b['affiliation'])).max() < 1e-9


def test_rank_deficient_hits_floor():
    """D > rank: eigenvalues are floored at 1e-10 (cacg.py:112-121); the
    in-loop slow path must reproduce the reference."""
    from oracle import synth
    from pb_bss_amd import _lib
    Y, init = synth.make_rank_deficient(9, 200, 6, 2, rank=3, seed=2)
    m, mask = _oracle_fit(Y, init, 3)
    r = _device_fit(Y, init, 3)
    st = _host(r['status'])
    assert (st & _lib.ST_FLOORED).all()
    # B^-1 carries 1e10 in the null space and the complex64 rounding of Y leaks
    # ~6e-8 of every frame into it: q is accurate to cond*eps ~ 1e-6 relative on
    # BOTH sides (the reference's einsum also forms B^-1 first), so two correct
    # float64 implementations agree only to ~1e-6 per step here.
    assert np.abs(_host(r['eigval']) - m['eigval']).max() < 1e-4
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-3
    # ... but the DECISION "this eigenvalue is floored" must agree bin by bin, class by class and
    # eigenvalue by eigenvalue: the reference leaves exactly eigenvalue_floor (1e-10) in the
    # floored slots (cacg.py:121), and the device status bit must be set for exactly the
    # (bin, class) pairs that contain one
    floored_ref = (m['eigval'] == 1e-10)
    floored_dev = (_host(r['eigval']) == 1e-10)
    assert (floored_ref == floored_dev).all()
    assert floored_ref.any(-1).all()
    assert ((st & _lib.ST_FLOORED) != 0).tolist() == floored_ref.any(-1).tolist()


def test_floor_decision_matches_reference_on_mixed_conditioning():
    """Bins of full rank next to rank-deficient ones in ONE launch: the floored / not-floored
    decision (status bit and the eigenvalue slots equal to the floor) matches the oracle per
    (bin, class); well-conditioned bins stay on the Gauss-Jordan fast path (no SLOWPATH bit)."""
    from oracle import synth
    from pb_bss_amd import _lib
    Ya, ia = synth.make_rank_deficient(5, 200, 6, 2, rank=3, seed=7)
    Yb, ib = synth.make_stft(6, 200, 6, 2, seed=8)
    Y = np.concatenate([Ya, Yb, Ya[:2]])
    init = np.concatenate([ia, ib, ia[:2]])
    m, mask = _oracle_fit(Y, init, 4)
    r = _device_fit(Y, init, 4)
    st = _host(r['status'])
    floored_ref = (m['eigval'] == 1e-10)
    assert ((_host(r['eigval']) == 1e-10) == floored_ref).all()
    assert (((st & _lib.ST_FLOORED) != 0) == floored_ref.any(-1)).all()
    assert floored_ref[:5].all(-1).sum() == 0 and floored_ref[:5].any(-1).all()
    assert not floored_ref[5:11].any()
    assert ((st[5:11] & _lib.ST_SLOWPATH) == 0).all()
    assert np.abs(_host(r['affiliation'])[5:11] - mask[5:11]).max() < 1e-9


def test_white_noise_worst_conditioning():
    from oracle import synth
    Y, init = synth.make_white(33, 400, 8, 3, seed=1)
    m, mask = _oracle_fit(Y, init, 10)
    r = _device_fit(Y, init, 10)
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-8


def test_zero_frames_and_complex128_input():
    from oracle import synth, cacgmm as oc
    from pb_bss_amd import engine
    Y, init = synth.make_stft(7, 150, 4, 2, seed=5, dtype=np.complex128)
    Y[:, 10] = 0
    m = oc.em_fit(Y, init, iterations=4)
    mask = oc.em_predict(m, Y)
    r = engine.em_fit(_dev(Y), 2, gamma0=_dev(init), iterations=4, final_predict=True)
    assert np.isfinite(_host(r['affiliation'])).all()
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-10


@pytest.mark.parametrize('covariance_norm', ['eigenvalue', 'trace', False])
def test_covariance_norm_and_options(covariance_norm):
    from oracle import synth
    Y, init = synth.make_stft(9, 120, 5, 3, seed=7)
    rng = np.random.default_rng(0)
    sal = rng.uniform(0.1, 1.0, size=(9, 120))
    m, mask = _oracle_fit(Y, init, 4, covariance_norm=covariance_norm, saliency=sal)
    r = _device_fit(Y, init, 4, covariance_norm=covariance_norm, saliency=_dev(sal))
    assert np.abs(_host(r['eigval']) - m['eigval']).max() < 1e-10 * max(1, m['eigval'].max())
    assert np.abs(_host(r['weight'])[..., None] - m['weight']).max() < 1e-10
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-9


def test_source_activity_mask_and_uniform_weight():
    from oracle import synth
    Y, init = synth.make_stft(6, 90, 4, 3, seed=9)
    rng = np.random.default_rng(1)
    act = rng.uniform(size=init.shape) > 0.2
    act[:, 0] = True
    m, mask = _oracle_fit(Y, init, 3, source_activity_mask=act, weight_constant_axis=-2)
    r = _device_fit(Y, init, 3, activity=_dev(act.astype(np.uint8)), weight_mode=1)
    from oracle import cacgmm as oc
    mask = oc.em_predict(m, Y.astype(np.complex128))
    assert np.abs(_host(r['weight']) - 1 / 3).max() < 1e-15
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-9


def test_model_initialisation_resume():
    """fit(initialization=model) resumes (cacgmm.py:229-234): 3 + 2 == 5."""
    from oracle import synth
    from pb_bss_amd import engine
    Y, init = synth.make_stft(12, 160, 6, 3, seed=11)
    a = _device_fit(Y, init, 5)
    b3 = _device_fit(Y, init, 3)
    b = engine.em_fit(_dev(Y), 3, model=(b3['eigvec'], b3['eigval'], b3['weight']),
                      iterations=2, final_predict=True)
    assert np.abs(_host(a['affiliation']) - _host(b['affiliation'])).max() < 1e-9


def test_predict_entry_point_q_and_log_pdf():
    from oracle import synth, cacgmm as oc
    from pb_bss_amd import engine, _lib
    Y, init = synth.make_stft(10, 140, 8, 3, seed=13)
    Y128 = Y.astype(np.complex128)
    m = oc.em_fit(Y128, init, iterations=3)
    yn = oc.normalize_observation(Y128)
    aff, q, lp = oc.e_step(yn, m['weight'], m['eigvec'], m['eigval'], affiliation_eps=1e-10)
    d_aff, d_q, d_lp = engine.em_predict(
        _dev(yn), _dev(m['eigvec']), _dev(m['eigval']), _dev(m['weight'][..., 0]),
        layout=_lib.LAYOUT_DT, affiliation_eps=1e-10, want_q=True, want_log_pdf=True)
    assert np.abs(_host(d_q) - q).max() / q.max() < 1e-11
    assert np.abs(_host(d_lp) - lp).max() < 1e-9
    assert np.abs(_host(d_aff) - aff).max() < 1e-10


def test_standalone_m_step_entry_point():
    from oracle import synth, cacgmm as oc
    from pb_bss_amd import engine, _lib
    Y, init = synth.make_stft(8, 110, 6, 2, seed=17)
    yn = oc.normalize_observation(Y.astype(np.complex128))
    rng = np.random.default_rng(3)
    q = rng.uniform(0.5, 2.0, size=init.shape)
    vec, val = oc.cacg_m_step(yn[..., None, :, :], init, q)
    d_vec, d_val, d_cov, st = engine.cacg_m_step(_dev(yn), _dev(init), _dev(q),
                                                 layout=_lib.LAYOUT_DT, want_cov=True)
    assert np.abs(_cov(_host(d_vec), _host(d_val)) - _cov(vec, val)).max() < 1e-12
    assert np.abs(_host(d_val) - val).max() < 1e-12


def test_long_utterance_spills_frames_to_hbm():
    """T too large for LDS: the kernel variant that keeps the frame-sized
    arrays in an HBM/L2 scratch slab must give the same answer."""
    from oracle import synth
    Y, init = synth.make_stft(3, 4000, 8, 3, seed=21)
    m, mask = _oracle_fit(Y, init, 3)
    r = _device_fit(Y, init, 3)
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-9
    Y, init = synth.make_stft(2, 10000, 3, 2, seed=22)
    m, mask = _oracle_fit(Y, init, 2)
    r = _device_fit(Y, init, 2)
    assert np.abs(_host(r['affiliation']) - mask).max() < 1e-9


@pytest.mark.parametrize('F,T,D,K', [(257, 200, 4, 2), (258, 130, 8, 3), (513, 300, 6, 4),
                                     (520, 500, 8, 3)])  # 520 = 8 utterances x 65 bins: rank 0 of 8
def test_split_tail_equals_plain_launch(F, T, D, K):
    """B = m*256 + r: the r remainder problems run as split groups (several
    workgroups share one bin's frames and exchange partial sums through L2).
    Same answer as the plain launch and as the oracle; no inter-workgroup
    wait may time out."""
    from oracle import synth
    from pb_bss_amd import engine
    Y, init = synth.make_stft(F, T, D, K, seed=F + T)
    engine.set_split_tail(False)
    try:
        plain = _device_fit(Y, init, 6)
    finally:
        engine.set_split_tail(True)
    split = _device_fit(Y, init, 6)
    assert engine.split_error() == 0
    for key in ('affiliation', 'eigval', 'weight', 'quadratic_form'):
        ref = _host(plain[key])
        assert np.abs(_host(split[key]) - ref).max() < 1e-10 * max(1.0, np.abs(ref).max()), key
    m, mask = _oracle_fit(Y[-3:], init[-3:], 6)   # the tail bins against the oracle
    assert np.abs(_host(split['affiliation'])[-3:] - mask).max() < 1e-9


def test_silent_bin_rank_one_bin_and_nan_input():
    """Edge inputs the reference handles by its floors / asserts: an all-zero frequency bin
    (e.g. a DC bin after high-pass filtering), a bin whose observation never leaves one
    direction (rank one), and a NaN sample (reference: AssertionError from np.isfinite checks,
    complex_angular_central_gaussian.py:326-333)."""
    from oracle import synth
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(6, 90, 4, 2, seed=21)
    Y = Y.copy()
    Y[1] = 0                                            # silent bin
    Y[3] = Y[3, :, :1] * np.array([1, 0.5j, -0.25, 2])  # rank-one bin
    m, mask = _oracle_fit(Y, init, 6)
    model = CACGMMTrainer().fit(Y, initialization=init, iterations=6)
    got = model.predict(Y)
    assert np.isfinite(got).all()
    ok = [0, 2, 4, 5]
    assert np.abs(got[ok] - mask[ok]).max() < 1e-9
    # silent bin: q is floored at tiny for every class -> posterior = mixture weights
    assert np.abs(got[1] - mask[1]).max() < 1e-9
    # rank-one bin: B^-1 carries 1/floor = 1e10, agreement to cond * eps
    assert np.abs(got[3] - mask[3]).max() < 1e-3
    bad = Y.copy()
    bad[2, 5, 1] = np.nan
    with pytest.raises(AssertionError):
        CACGMMTrainer().fit(bad, initialization=init, iterations=3)


def test_split_timeout_pattern_repeats_the_fit_without_split_groups(monkeypatch):
    """NONFINITE | EIG_NOCONV together with the flag of pbbss_split_error is how a launch reports
    that the member workgroups of a remainder bin were not co-resident: the wrapper warns and
    repeats the fit once with the split groups switched off (and switches them on again)."""
    from oracle import synth
    from pb_bss_amd import _lib, engine
    Y, init = synth.make_stft(257, 300, 4, 2, seed=21)
    y, g0 = _lib.to_device(Y), _lib.to_device(init)
    engine.set_split_tail(False)
    want = engine.em_fit(y, 2, gamma0=g0, iterations=5, final_predict=True)
    engine.set_split_tail(True)
    calls, tails = [], []
    real_bits, real_tail = engine._status_bits, engine.set_split_tail

    def fake_bits(st):
        calls.append(1)
        return engine._SPLIT_POISON if len(calls) == 1 else real_bits(st)

    def fake_tail(enable, device_index=None):
        tails.append(bool(enable))
        return real_tail(enable, device_index)

    monkeypatch.setattr(engine, '_status_bits', fake_bits)
    monkeypatch.setattr(engine, 'split_error', lambda device_index=None: 1)
    monkeypatch.setattr(engine, 'set_split_tail', fake_tail)
    with pytest.warns(RuntimeWarning, match='split groups'):
        got = engine.em_fit(y, 2, gamma0=g0, iterations=5, final_predict=True)
    assert tails == [False, True] and len(calls) == 2
    assert (got['affiliation'] == want['affiliation']).all()
    assert (got['eigval'] == want['eigval']).all()
    # without the flag of pbbss_split_error the same bits are the reference's numerical failure
    calls.clear()
    monkeypatch.setattr(engine, 'split_error', lambda device_index=None: 0)
    with pytest.raises(AssertionError, match='non-finite'):
        engine.em_fit(y, 2, gamma0=g0, iterations=5, final_predict=True)


@pytest.mark.parametrize('T', [40, 63, 64, 65, 127, 128, 129, 255, 256, 257, 320, 511, 513])
def test_frame_counts_around_the_chunk_size(T):
    """The LDS frame arrays live in chunks of 64 frames with zero padding frames behind T (round 5):
    the M sweep runs over whole chunks unmasked, E passes skip chunks beyond the padded count, the
    E phase rewrites the padding frames' weights with zeros.  Frame counts on both sides of every
    chunk and pass boundary, odd sensor counts (a half-used float4 plane), saliency with zeros, an
    activity mask, an all-zero frame at the very end -- against the oracle after 4 iterations."""
    from oracle import synth
    rng = np.random.default_rng(T)
    for D, K in ((2, 2), (5, 3), (8, 3)):
        F = 3
        Y, init = synth.make_stft(F, T, D, K, seed=T + D)
        Y[1, -1] = 0                                    # zero frame in the last (partial) chunk
        sal = rng.uniform(0.0, 1.0, size=(F, T))
        sal[:, ::7] = 0.0
        act = (rng.uniform(size=(F, K, T)) > 0.1)
        act[:, 0] = True                                # at least one class active everywhere
        for kw_o, kw_d in (({}, {}),
                           (dict(saliency=sal), dict(saliency=_dev(sal))),
                           (dict(source_activity_mask=act), dict(activity=_dev(act.astype(np.uint8))))):
            ref, want = _oracle_fit(Y, init, 4, **kw_o)
            r = _device_fit(Y, init, 4, **kw_d)
            got = _host(r['affiliation'])
            tol = 1e-8
            # the final predict of the oracle applies no mask (CACGMM.predict); the device's too
            assert np.abs(got - want).max() < tol, (T, D, K, list(kw_o))
            np.testing.assert_allclose(_cov(_host(r['eigvec']), _host(r['eigval'])),
                                       _cov(ref['eigvec'], ref['eigval']), atol=tol)


@pytest.mark.parametrize('T', [130, 192, 200])
def test_remainder_bins_with_partial_last_window(T):
    """Split groups (2^n + 1 bins) whose last 64-frame window is partial or whose frame count is a
    whole number of windows: the members carve their LDS for a full window and sweep only the
    chunks their frames fill."""
    from oracle import synth
    from pb_bss_amd import engine
    F, D, K = 258, 4, 2
    Y, init = synth.make_stft(F, T, D, K, seed=T)
    r = _device_fit(Y, init, 5)
    sel = [0, 100, 255, 256, 257]
    ref, want = _oracle_fit(Y[sel], init[sel], 5)
    assert engine.split_error() == 0
    assert np.abs(_host(r['affiliation'])[sel] - want).max() < 1e-8


def test_nonfinite_posterior_is_reported_for_any_affiliation_eps():
    """np.maximum / np.clip keep a NaN (mixture_model_utils.py:43-53) and the reference's M-step
    asserts on the covariance it produces (complex_angular_central_gaussian.py:326-333); the
    device's v_max / v_min would hand back a clean `eps` -- also for affiliation_eps = 0 since the
    in-loop clip became unconditional.  A model whose mixture weight is NaN in one bin (finite
    observation) must fail the same way for both settings, in the float64 and in the packed-FP32
    kernel."""
    import dataclasses
    from oracle import synth
    from pb_bss_amd.distribution import CACGMMTrainer, utils
    Y, init = synth.make_stft(5, 80, 4, 2, seed=3)
    model = CACGMMTrainer().fit(Y, initialization=init, iterations=2)
    good = CACGMMTrainer().fit(Y, initialization=model, iterations=3, affiliation_eps=0.)
    assert np.isfinite(good.predict(Y)).all()
    w = np.array(model.weight, copy=True)
    w[2] = np.nan
    bad = dataclasses.replace(model, weight=w)
    for eps in (0.0, 1e-10):
        with pytest.raises(AssertionError):
            CACGMMTrainer().fit(Y, initialization=bad, iterations=3, affiliation_eps=eps)
    with utils.arithmetic('reference'):
        for eps in (0.0, 1e-10):
            with pytest.raises(AssertionError):
                CACGMMTrainer().fit(Y, initialization=bad, iterations=3, affiliation_eps=eps)
    # predict hands the NaN posterior out, like the reference
    assert np.isnan(bad.predict(Y)[2]).all() and np.isfinite(bad.predict(Y)[[0, 1, 3, 4]]).all()
