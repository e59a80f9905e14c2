"""GPU: REAL time-outs of the inter-workgroup protocols, provoked with the test knob
pbbss_set_spin_limit (a bounded wait gives up after ONE poll although its peers are on the chip).
Round 3 tested these paths by patching status words; here the kernels themselves run out of
patience: the split groups of a remainder bin (cACGMM and Watson kernels), the cooperative
shared-weight launch and the DHTV teams must each (i) never hang, (ii) report the time-out the
way the Python layer expects -- status poison + pbbss_split_error, DHTV status -- and (iii) end in
the same result as the path without inter-workgroup waits, with a RuntimeWarning."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def one_poll():
    from pb_bss_amd import engine
    engine.split_reset()
    engine.set_spin_limit(1)
    try:
        yield
    finally:
        engine.set_spin_limit(0)
        engine.split_reset()
        engine.set_split_tail(True)


def test_split_groups_really_time_out_and_the_fit_is_repeated(one_poll):
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.testing import synth
    Y, init = synth.make_stft(257, 512, 4, 2, seed=31)       # 257 = 256 + 1 bins: 8 split windows
    y, g0 = _lib.to_device(Y), _lib.to_device(init)
    engine.set_split_tail(False)
    want = engine.em_fit(y, 2, gamma0=g0, iterations=8, final_predict=True)
    engine.set_split_tail(True)
    # the raw launch: the members give up, the status words of the split problem carry the
    # poison pattern and the handle's flag is up
    raw = engine.em_fit(y, 2, gamma0=g0, iterations=8, final_predict=True, check_status=False)
    poison = _lib.ST_NONFINITE | _lib.ST_EIG_NOCONV
    st = _lib.to_host(raw['status'])
    assert (st[256] & poison == poison).all() and (st[:256] & poison == 0).all()
    assert engine.split_error() == 1
    engine.split_reset()
    assert engine.split_error() == 0
    with pytest.warns(RuntimeWarning, match='split groups'):
        got = engine.em_fit(y, 2, gamma0=g0, iterations=8, final_predict=True)
    assert engine.split_tail() is True                        # the caller's setting is back
    assert engine.split_error() == 0                          # the report was consumed
    assert (got['affiliation'] == want['affiliation']).all()
    assert (got['eigval'] == want['eigval']).all()
    # with the waits back to normal the very same handle runs its split groups again
    engine.set_spin_limit(0)
    ok = engine.em_fit(y, 2, gamma0=g0, iterations=8, final_predict=True)
    assert np.abs(_lib.to_host(ok['affiliation']) - _lib.to_host(want['affiliation'])).max() < 1e-9


def test_watson_split_groups_really_time_out(one_poll):
    from pb_bss_amd.distribution import CWMMTrainer
    from pb_bss_amd import engine
    from pb_bss_amd.testing import synth
    from oracle import cwmm as ow
    Y, init = synth.make_stft(257, 512, 4, 2, seed=32)
    with pytest.warns(RuntimeWarning, match='split groups'):
        masks = CWMMTrainer().fit_predict(Y, initialization=init, iterations=6)
    Y128 = Y.astype(np.complex128)
    ref = ow.cwmm_predict(ow.cwmm_fit(Y128, init, iterations=6), Y128)
    assert np.abs(masks - ref).max() < 1e-7
    assert engine.split_error() == 0


def test_cooperative_shared_weight_launch_really_times_out(one_poll):
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.testing import synth
    from oracle import cacgmm as oc
    # 257 bins on 256 compute units, two workgroups of this size per unit at most: the unit that
    # hosts two runs its phases ~1.4x slower, the other 255 workgroups wait for it at every grid
    # barrier for several microseconds -- far longer than the two polls they are allowed
    Y, init = synth.make_stft(257, 700, 8, 3, seed=33)
    with pytest.warns(RuntimeWarning, match='cooperative shared-weight launch timed out'):
        masks = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=5,
                                            weight_constant_axis=(-3, -1))
    Y128 = Y.astype(np.complex128)
    ref = oc.em_predict(oc.em_fit(Y128, init, iterations=5, weight_constant_axis=(-3, -1)), Y128)
    assert np.abs(masks - ref).max() < 1e-9


def test_dhtv_team_really_times_out(one_poll):
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from oracle import permutation_alignment as op
    rng = np.random.default_rng(34)
    K, F, T = 3, 257, 300
    act = rng.uniform(size=(K, 1, T)) ** 4
    mask = act * rng.uniform(0.5, 1.0, size=(K, F, T)) + 0.05 * rng.uniform(size=(K, F, T))
    mask /= mask.sum(0, keepdims=True)
    for f in range(F):
        mask[:, f] = mask[rng.permutation(K), f]
    solver = DHTVPermutationAlignment.from_stft_size(512)
    with pytest.warns(RuntimeWarning, match='co-resident'):
        got = solver.calculate_mapping(mask)
    want = op.dhtv_calculate_mapping(mask, op.alignment_plan(512, **op.PRESETS[512]))
    assert np.array_equal(got, want)


def test_inline_dhtv_team_time_out_repeats_the_fit_on_one_workgroup(one_poll):
    """The inline aligner of CACGMMTrainer.fit reads its status words once after the loop: a team
    time-out in any iteration makes the whole loop run again on the one-workgroup kernel."""
    from pb_bss_amd import engine
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from pb_bss_amd.testing import synth
    Y, init = synth.make_stft(257, 300, 4, 3, seed=35)
    solver = DHTVPermutationAlignment.from_stft_size(512)
    kw = dict(initialization=init, iterations=3, weight_constant_axis=(-3,),
              inline_permutation_aligner=solver)
    with pytest.warns(RuntimeWarning, match='co-resident'):
        got = CACGMMTrainer().fit(Y, **kw)
    engine.set_spin_limit(0)
    want = CACGMMTrainer().fit(Y, **kw)
    # (the remainder bin's sums are added in another order without split groups: not bit-equal)
    assert np.abs(got.weight - want.weight).max() < 1e-10
    assert np.abs(got.cacg.covariance_eigenvalues - want.cacg.covariance_eigenvalues).max() < 1e-8
