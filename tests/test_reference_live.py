"""CPU, build container only (skipped where /root/reference is absent): the committed golden
fixtures are what the UNMODIFIED reference produces today, and the NumPy oracle agrees with the
live reference on freshly drawn inputs -- i.e. the pin of the oracle can be re-derived, not just
trusted.  The reference is imported through oracle/refshim.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.needs_reference
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ref():
    import warnings
    from oracle import refshim
    refshim.load()
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    import pb_bss.distribution as dist
    import pb_bss.extraction.beamformer as bf
    import pb_bss.permutation_alignment as pa
    return dist, bf, pa


def test_committed_fixture_equals_live_reference(ref):
    dist, bf, _ = ref
    g = np.load(os.path.join(GOLDEN, 'cacgmm_f9_t120_d8_k3.npz'))
    model = dist.CACGMMTrainer().fit(g['Y'].astype(np.complex128), initialization=g['init'],
                                     iterations=int(g['iterations']))
    np.testing.assert_allclose(model.predict(g['Y'].astype(np.complex128)), g['affiliation'],
                               atol=1e-12)
    g = np.load(os.path.join(GOLDEN, 'beamformer_extra_f19_d5.npz'))
    np.testing.assert_allclose(bf.get_lcmv_vector(g['atf'], [1, 0], g['noise']), g['lcmv_10'],
                               atol=1e-12)
    np.testing.assert_allclose(bf.phase_correction(g['wb']), g['phase_3d'], atol=1e-14)


def test_oracle_equals_live_reference_on_fresh_inputs(ref):
    dist, bf, pa = ref
    from oracle import beamformer as ob, cacgmm as oc, embed as oe, permutation_alignment as op, synth
    Y, e, init = synth.make_joint(7, 90, 5, 2, 8, seed=1234)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    m = dist.CACGMMTrainer().fit(Y128, initialization=init, iterations=6)
    o = oc.em_fit(Y128, init, iterations=6)
    np.testing.assert_allclose(oc.em_predict(o, Y128), m.predict(Y128), atol=1e-10)
    mj = dist.GCACGMMTrainer().fit(Y128, e64, initialization=init, iterations=4)
    oj = oe.joint_fit('gaussian', Y128, e64, init, 4)
    np.testing.assert_allclose(oe.joint_model_predict(oj, Y128, e64), mj.predict(Y128, e64),
                               atol=1e-10)
    masks = m.predict(Y128)
    X = Y128.transpose(0, 2, 1)
    psd = bf.get_power_spectral_density_matrix(X, masks)
    np.testing.assert_allclose(ob.psd(X, masks), psd, atol=1e-13)
    w = bf.get_mvdr_vector_souden(psd[:, 0], psd[:, 1], ref_channel=0)
    np.testing.assert_allclose(ob.mvdr_souden(psd[:, 0], psd[:, 1], ref_channel=0), w, atol=1e-12)
    solver = pa.DHTVPermutationAlignment(stft_size=12, segment_start=2, segment_width=3,
                                         segment_shift=1, main_iterations=5, sub_iterations=2)
    kft = np.ascontiguousarray(masks.transpose(1, 0, 2))
    np.testing.assert_array_equal(
        op.dhtv_calculate_mapping(kft, solver.alignment_plan), solver.calculate_mapping(kft))
    for metric in ('cos', 'multiply', 'euclidean'):
        np.testing.assert_array_equal(
            op.greedy_calculate_mapping(kft, metric),
            pa.GreedyPermutationAlignment(similarity_metric=metric).calculate_mapping(kft))
        shuffled = np.ascontiguousarray(kft[::-1])
        for alg in ('greedy', 'optimal'):
            np.testing.assert_array_equal(
                op.oracle_calculate_mapping(shuffled, kft, metric, alg),
                pa.OraclePermutationAlignment(metric, alg).calculate_mapping(shuffled, kft))


def test_oracle_gmm_equals_live_reference_on_fresh_inputs(ref):
    """GMMTrainer (covariance_type 'full' -- its default -- and 'spherical') and the Gaussian
    single fits of the unmodified reference against the NumPy oracle on freshly drawn data."""
    dist, _, _ = ref
    from oracle import embed as oe
    rng = np.random.default_rng(4321)
    N, E, K = 700, 9, 3
    centers = rng.normal(size=(K, E)) * 2
    y = centers[rng.integers(K, size=N)] + rng.normal(size=(N, E))
    init = rng.uniform(size=(K, N))
    init /= init.sum(0, keepdims=True)
    sal = rng.uniform(0.2, 1.0, size=N)
    for ct in ('full', 'spherical'):
        m = dist.GMMTrainer().fit(y, initialization=init, iterations=5, saliency=sal,
                                  covariance_type=ct)
        o = oe.gmm_fit(y, init, 5, saliency=sal, covariance_type=ct)
        np.testing.assert_allclose(o['mean'], m.gaussian.mean, atol=1e-11)
        np.testing.assert_allclose(o['covariance'], m.gaussian.covariance, atol=1e-11)
        np.testing.assert_allclose(o['weight'], m.weight, atol=1e-12)
        np.testing.assert_allclose(oe.gmm_predict(o, y, ct), m.predict(y), atol=1e-10)
    w = rng.uniform(size=(K, N))
    g = dist.GaussianTrainer()._fit(y[None], saliency=w, covariance_type='full')
    om, oc = oe.gaussian_fit(y[None], w, 'full')
    np.testing.assert_allclose(om, g.mean, atol=1e-12)
    np.testing.assert_allclose(oc, g.covariance, atol=1e-12)
    np.testing.assert_allclose(oe.gaussian_log_pdf(y[None], om, oc, 'full'), g.log_pdf(y[None]),
                               rtol=1e-10, atol=1e-9)


def test_oracle_mixture_weight_axes_equal_live_reference(ref):
    """The options the step-wise device loop serves (tests/test_gpu_embed_stepwise.py compares it
    with the oracle): weight_constant_axis sets over an independent axis for VMFMM / GMM and
    covariance_type='diagonal' -- oracle == unmodified reference on fresh inputs."""
    from pb_bss.distribution import GMMTrainer, VMFMMTrainer
    from oracle import embed as oe
    rng = np.random.default_rng(11)
    y = rng.standard_normal((3, 120, 5)) + rng.integers(0, 3, size=(3, 120, 1)) * 2.0
    init = rng.uniform(size=(3, 3, 120))
    init /= init.sum(axis=1, keepdims=True)
    sal = rng.uniform(0.2, 1.0, size=(3, 120))
    for axis in ((-3,), (-3, -1)):
        m = VMFMMTrainer().fit(y, initialization=init, iterations=3, saliency=sal,
                               weight_constant_axis=axis)
        o = oe.vmfmm_fit(y, init, iterations=3, saliency=sal, weight_constant_axis=axis)
        assert np.abs(m.weight - o['weight']).max() < 1e-12
        assert np.abs(m.vmf.mean - o['mean']).max() < 1e-12
        for cov in ('full',):  # SphericalGaussian.log_pdf does not broadcast over independent
            # axes in the reference (gaussian.py:110-113): 'spherical' is pinned on flat data only
            g = GMMTrainer().fit(y, initialization=init, iterations=3, saliency=sal,
                                 weight_constant_axis=axis, covariance_type=cov)
            og = oe.gmm_fit(y, init, iterations=3, saliency=sal, weight_constant_axis=axis,
                            covariance_type=cov)
            assert np.abs(g.weight - og['weight']).max() < 1e-12
            assert np.abs(g.gaussian.covariance - og['covariance']).max() < 1e-10
    for axis in ((-1,), (-2,), -2):
        g = GMMTrainer().fit(y[0], initialization=init[0], iterations=3, saliency=sal[0],
                             weight_constant_axis=axis, covariance_type='diagonal')
        og = oe.gmm_fit(y[0], init[0], iterations=3, saliency=sal[0], weight_constant_axis=axis,
                        covariance_type='diagonal')
        assert np.abs(g.gaussian.mean - og['mean']).max() < 1e-12
        assert np.abs(g.gaussian.covariance - og['covariance']).max() < 1e-12
        assert np.abs(g.predict(y[0]) - oe.gmm_predict(og, y[0], 'diagonal')).max() < 1e-10


def test_gev_use_eig_fixture_equals_reference_cython_modules():
    """The reference's two native files compile out of tree (oracle/refshim.py:build_cython ->
    oracle/_ref/); with them injected, `get_gev_vector(use_eig=True)` runs c_eig.pyx (zggev) and
    the default path get_gev_vector.pyx (zhegvd).  The committed fixture is what they produce
    today, and the oracle agrees with them on fresh pencils.  Runs in a subprocess: the modules
    must be injected before pb_bss.extraction.beamformer is imported."""
    import subprocess
    import sys
    code = r'''
import os, sys, warnings
import numpy as np
warnings.simplefilter('ignore')
from oracle import refshim, beamformer as ob
refshim.load_cython()
import pb_bss.extraction.beamformer as bf
assert bf.c_gev_available and bf.c_eig_available
g = np.load(os.path.join('tests', 'golden', 'gev_use_eig.npz'))
cos = lambda a, b: np.abs((a.conj() * b).sum(-1)) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)
for tag in ('hpd_d6', 'indef_d5', 'general_d6'):
    w = bf.get_gev_vector(g[tag + '_target'], g[tag + '_noise'], use_eig=True)
    assert np.abs(cos(w, g[tag + '_w_cython']) - 1).max() < 1e-12, tag
rng = np.random.default_rng(99)
A = rng.standard_normal((6, 5, 5)) + 1j * rng.standard_normal((6, 5, 5))
B = rng.standard_normal((6, 5, 5)) + 1j * rng.standard_normal((6, 5, 5)) + 2 * np.eye(5)
assert np.abs(cos(bf.get_gev_vector(A, B, use_eig=True), ob.gev_vector_eig(A, B)) - 1).max() < 1e-12
X = rng.standard_normal((6, 5, 20)) + 1j * rng.standard_normal((6, 5, 20))
P = X @ X.conj().swapaxes(-1, -2)
Q = P + np.eye(5)
assert np.abs(cos(bf.get_gev_vector(P, Q), ob.gev_vector(P, Q)) - 1).max() < 1e-12   # zhegvd module
print('OK')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and 'OK' in out.stdout, out.stderr[-2000:]
