"""GPU: the step-wise device loop of the real-embedding mixtures (pb_bss_amd/distribution/
_embed_stepwise.py) -- weight_constant_axis sets beyond the fused loops, frame-varying weights and
covariance_type='diagonal' -- and its new C entry point pbbss_log_pdf_to_affiliation, against the
NumPy oracle (oracle/embed.py, pinned to the live reference for these options by
tests/test_reference_live.py::test_oracle_mixture_weight_axes_equal_live_reference)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(B, N, E, K, seed):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((B, K, E)) * 2.0
    lab = rng.integers(0, K, size=(B, N))
    y = mu[np.arange(B)[:, None], lab] + 0.6 * rng.standard_normal((B, N, E))
    init = rng.uniform(size=(B, K, N))
    init /= init.sum(axis=1, keepdims=True)
    sal = rng.uniform(0.2, 1.0, size=(B, N))
    return y, init, sal


@pytest.mark.parametrize('shape', ['bk1', 'k1', '1kn', 'b1n', 'bkn'])
def test_log_pdf_to_affiliation_kernel(shape):
    from oracle import cacgmm as oc
    from pb_bss_amd import _lib, engine
    rng = np.random.default_rng(5)
    B, K, N = 3, 4, 700
    lp = rng.standard_normal((B, K, N)) * 30.0
    w = {'bk1': rng.uniform(size=(B, K, 1)), 'k1': rng.uniform(size=(K, 1)),
         '1kn': rng.uniform(size=(1, K, N)), 'b1n': rng.uniform(size=(B, 1, N)),
         'bkn': rng.uniform(size=(B, K, N))}[shape]
    act = rng.uniform(size=(B, K, N)) > 0.2
    for mask, eps in ((None, 0.0), (act, 1e-10)):
        got = engine.log_pdf_to_affiliation(
            _lib.to_device(lp), _lib.to_device(w),
            activity=None if mask is None else _lib.to_device(mask.astype(np.uint8)),
            affiliation_eps=eps)
        ref = oc.log_pdf_to_affiliation(w, lp, source_activity_mask=mask, affiliation_eps=eps)
        assert np.abs(_lib.to_host(got) - ref).max() < 1e-14


@pytest.mark.parametrize('axis', [(-3,), (-3, -1), [-1, -3]])
@pytest.mark.parametrize('with_sal', [False, True])
def test_vmfmm_weights_shared_over_an_independent_axis(axis, with_sal):
    from oracle import embed as oe
    from pb_bss_amd.distribution import VMFMMTrainer
    y, init, sal = _data(4, 300, 8, 3, seed=1)
    sal = sal if with_sal else None
    model = VMFMMTrainer().fit(y, initialization=init, iterations=5, saliency=sal,
                               weight_constant_axis=axis)
    ref = oe.vmfmm_fit(y, init, iterations=5, saliency=sal, weight_constant_axis=tuple(axis))
    assert model.weight.shape == ref['weight'].shape
    assert np.abs(model.weight - ref['weight']).max() < 1e-10
    assert np.abs(model.vmf.mean - ref['mean']).max() < 1e-9
    assert np.abs(model.vmf.concentration - ref['concentration']).max() < 1e-7 * np.abs(
        ref['concentration']).max()
    assert np.abs(model.predict(y) - oe.vmfmm_predict(ref, y)).max() < 1e-8


@pytest.mark.parametrize('cov,axis', [('spherical', (-3, -1)), ('spherical', (-3,)),
                                      ('full', (-3, -1)), ('full', (-3,))])
def test_gmm_weights_shared_over_an_independent_axis(cov, axis):
    from oracle import embed as oe
    from pb_bss_amd.distribution import GMMTrainer
    y, init, sal = _data(3, 400, 6, 3, seed=2)
    model = GMMTrainer().fit(y, initialization=init, iterations=4, saliency=sal,
                             weight_constant_axis=axis, covariance_type=cov)
    ref = oe.gmm_fit(y, init, iterations=4, saliency=sal, weight_constant_axis=axis,
                     covariance_type=cov)
    assert model.weight.shape == ref['weight'].shape
    assert np.abs(model.weight - ref['weight']).max() < 1e-10
    assert np.abs(model.gaussian.mean - ref['mean']).max() < 1e-9
    assert np.abs(model.gaussian.covariance - ref['covariance']).max() < 1e-9
    assert np.abs(model.predict(y) - oe.gmm_predict(ref, y, cov)).max() < 1e-8


@pytest.mark.parametrize('axis', [(-1,), (-2,), -2])
def test_gmm_diagonal_covariances(axis):
    """covariance_type='diagonal' exactly as the reference evaluates it (its log-pdf feeds the
    (K, D) precision array to einsum as ONE K x D matrix, gaussian.py:87-91) on a flat mixture."""
    from oracle import embed as oe
    from pb_bss_amd.distribution import GMMTrainer
    y, init, sal = _data(1, 500, 5, 3, seed=3)
    y, init, sal = y[0], init[0], sal[0]
    tr = GMMTrainer()
    model = tr.fit(y, initialization=init, iterations=4, saliency=sal, weight_constant_axis=axis,
                   covariance_type='diagonal')
    ref = oe.gmm_fit(y, init, iterations=4, saliency=sal, weight_constant_axis=axis,
                     covariance_type='diagonal')
    assert type(model.gaussian).__name__ == 'DiagonalGaussian'
    assert np.abs(model.gaussian.mean - ref['mean']).max() < 1e-9
    assert np.abs(model.gaussian.covariance - ref['covariance']).max() < 1e-9
    assert np.abs(np.asarray(model.weight) - ref['weight']).max() < 1e-10
    assert np.abs(model.predict(y) - oe.gmm_predict(ref, y, 'diagonal')).max() < 1e-8
    got = tr.fit_predict(y, initialization=init, iterations=4, saliency=sal,
                         weight_constant_axis=axis, covariance_type='diagonal')
    assert np.abs(got - oe.gmm_predict(ref, y, 'diagonal')).max() < 1e-8
