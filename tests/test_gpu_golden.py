"""GPU: the drop-in Python API (pb_bss_amd.distribution / .extraction, which
call the HIP library through the C ABI) against vectors produced by the REAL
reference (tests/golden, oracle/make_golden.py).  The device receives the
complex64 observation; the reference had its exact complex128 upcast."""
import ast
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def cos_sim(a, b):
    return np.abs(np.einsum('...d,...d->...', a.conj(), b)) / (
        np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


CACGMM_CASES = sorted(os.path.basename(p)[:-4] for p in
                      glob.glob(os.path.join(GOLDEN, 'cacgmm_f*.npz')))


@pytest.mark.parametrize('name', CACGMM_CASES)
def test_trainer_matches_reference(name):
    from pb_bss_amd.distribution import CACGMMTrainer
    g = load(name)
    kw = ast.literal_eval(str(g['kwargs']))
    model = CACGMMTrainer().fit(g['Y'], initialization=g['init'],
                                iterations=int(g['iterations']), **kw)
    assert model.weight.shape == g['weight'].shape
    assert model.cacg.covariance_eigenvectors.shape == g['eigvec'].shape
    assert np.abs(model.weight - g['weight']).max() < 1e-9
    assert np.abs(model.cacg.covariance - g['covariance']).max() < 1e-8
    assert np.abs(model.cacg.covariance_eigenvalues - g['eigval']).max() < 1e-8
    aff, q = model.predict(g['Y'], return_quadratic_form=True)
    assert aff.dtype == np.float64 and aff.shape == g['affiliation'].shape
    assert np.abs(aff - g['affiliation']).max() < 1e-8
    assert np.abs(q - g['quadratic_form']).max() / g['quadratic_form'].max() < 1e-8
    ll = model.log_likelihood(g['Y'])
    assert abs(ll - g['log_likelihood']) < 1e-7 * abs(g['log_likelihood'])


def test_saliency_zero_frame_rank_deficient_batch():
    from pb_bss_amd.distribution import CACGMMTrainer
    g = load('cacgmm_saliency_zero_frame')
    m = CACGMMTrainer().fit(g['Y'], initialization=g['init'],
                            iterations=int(g['iterations']), saliency=g['saliency'])
    assert np.abs(m.weight - g['weight']).max() < 1e-10
    assert np.abs(m.predict(g['Y']) - g['affiliation']).max() < 1e-9
    g = load('cacgmm_rank_deficient')
    m = CACGMMTrainer().fit(g['Y'], initialization=g['init'], iterations=int(g['iterations']))
    assert (m.cacg.covariance_eigenvalues.min(axis=-1) == 1e-10).all()
    assert np.abs(m.cacg.covariance_eigenvalues - g['eigval']).max() < 1e-4  # cond*eps, see test_gpu_em
    g = load('cacgmm_batch_axis')
    m = CACGMMTrainer().fit(g['Y'], initialization=g['init'], iterations=int(g['iterations']))
    assert m.cacg.covariance_eigenvectors.shape == (2, 3, 2, 4, 4)
    assert m.weight.shape == g['weight'].shape
    assert np.abs(m.predict(g['Y']) - g['affiliation']).max() < 1e-9


def test_cacg_doctest_and_m_step():
    from pb_bss_amd.distribution import (ComplexAngularCentralGaussian,
                                         ComplexAngularCentralGaussianTrainer,
                                         normalize_observation)
    g = load('cacg_fit_doctest')
    m = ComplexAngularCentralGaussianTrainer()._fit(y=g['y'], saliency=None,
                                                    quadratic_form=g['quadratic_form'])
    assert m.covariance_eigenvalues.shape == (2, 3)
    assert np.allclose(m.covariance_eigenvalues, [[1e-10, 1, 1], [1e-10, 1, 1]], atol=1e-15)
    assert np.abs(m.covariance - g['covariance']).max() < 1e-14
    g = load('cacg_m_step_log_pdf')
    yn = normalize_observation(g['Y'])
    assert yn.shape == g['yn'].shape and np.abs(yn - g['yn']).max() < 1e-15
    m = ComplexAngularCentralGaussianTrainer()._fit(
        y=yn[:, None], saliency=g['saliency'], quadratic_form=g['quadratic_form'])
    assert m.covariance_eigenvectors.shape == g['eigvec'].shape
    assert np.abs(m.covariance - g['covariance']).max() < 1e-12
    ref = ComplexAngularCentralGaussian(g['eigvec'], g['eigval'])
    lp, q = ref._log_pdf(yn[:, None])
    assert lp.shape == g['log_pdf'].shape
    assert np.abs(lp - g['log_pdf']).max() < 1e-9
    assert np.abs(q - g['q_out']).max() / g['q_out'].max() < 1e-11
    g = load('cacg_trainer_fit')
    m = ComplexAngularCentralGaussianTrainer().fit(g['Y'], iterations=5)
    assert np.abs(m.covariance - g['covariance']).max() < 1e-10
    assert np.abs(m.log_pdf(g['Y']) - g['log_pdf']).max() < 1e-9
    # from_covariance: eigh + normalisation + floor
    fc = ComplexAngularCentralGaussian.from_covariance(g['covariance'], eigenvalue_floor=1e-10)
    assert np.abs(fc.covariance - g['covariance'] / np.linalg.eigvalsh(g['covariance']).max()).max() < 1e-12


def test_beamformer_api_matches_reference():
    from pb_bss_amd import extraction as ex
    g = load('beamformer_f17_d6')
    X, mask = g['X'], g['mask']
    assert np.abs(ex.get_power_spectral_density_matrix(X, mask) - g['psd']).max() < 1e-12
    assert np.abs(ex.get_power_spectral_density_matrix(X, mask, normalize=False)
                  - g['psd_nonorm']).max() < 1e-10
    assert np.abs(ex.get_power_spectral_density_matrix(X) - g['psd_plain']).max() < 1e-12
    assert np.abs(ex.get_power_spectral_density_matrix(X, mask[:, 0]) - g['psd_2d']).max() < 1e-12
    p0 = ex.get_power_spectral_density_matrix(X, mask.transpose(1, 0, 2), source_dim=0)
    assert p0.shape == g['psd_src0'].shape and np.abs(p0 - g['psd_src0']).max() < 1e-12
    t, n = g['target'], g['noise']
    assert np.abs(cos_sim(ex.get_gev_vector(t, n), g['gev']) - 1).max() < 1e-10
    assert np.abs(cos_sim(ex.get_pca_vector(t), g['pca']) - 1).max() < 1e-10
    w, ref = ex.get_mvdr_vector_souden(t, n, return_ref_channel=True)
    assert ref == int(g['mvdr_souden_ref']) == ex.get_optimal_reference_channel(
        np.linalg.solve(n, t), t, n)
    assert np.abs(w - g['mvdr_souden']).max() < 1e-10
    assert np.abs(ex.get_mvdr_vector_souden(t, n, ref_channel=1) - g['mvdr_souden_ch1']).max() < 1e-10
    assert np.abs(ex.get_mvdr_vector(g['pca'], n) - g['mvdr']).max() < 1e-9
    assert np.abs(ex.blind_analytic_normalization(g['gev'], n) - g['ban']).max() < 1e-11
    assert np.abs(ex.apply_beamforming_vector(g['mvdr_souden'], X) - g['applied']).max() < 1e-11
    assert np.abs(ex.get_wmwf_vector(t, n) - g['wmwf']).max() < 1e-10
    assert np.abs(ex.get_wmwf_vector(t, n, reference_channel=2, distortion_weight=3.)
                  - g['wmwf_mu3_ch2']).max() < 1e-10
    assert np.abs(ex.get_wmwf_vector(t, n, reference_channel=0, distortion_weight='frequency_dependent')
                  - g['wmwf_freqdep']).max() < 1e-10
    sel = np.zeros(6); sel[1] = 1
    assert np.abs(ex.get_wmwf_vector(t, n, channel_selection_vector=sel)
                  - ex.get_wmwf_vector(t, n, reference_channel=1)).max() < 1e-15
    for key in g.files:
        if not key.startswith('bf__'):
            continue
        name = key[4:].replace('__', '+')
        w = ex.get_bf_vector(name, t, n)
        assert w.shape == g[key].shape, name
        assert np.abs(cos_sim(w, g[key]) - 1).max() < 1e-9, name
        assert np.abs(np.linalg.norm(w, axis=-1) - np.linalg.norm(g[key], axis=-1)).max() \
            < 1e-8 * np.linalg.norm(g[key], axis=-1).max(), name


def test_mvdr_souden_known_answer_and_stable_solve():
    from pb_bss_amd import extraction as ex
    g = load('mvdr_souden_kat')
    w, = ex.get_mvdr_vector_souden(g['pxx'][None], g['pnn'][None])
    # tests/test_extraction/test_beamformer.py:205-209 compares the repr
    assert np.allclose(w.real, [0.03311258, 0.03311258, 0.99337748], atol=5e-9)
    assert np.abs(w.imag).max() == 0
    X = ex.stable_solve(g['solve_A'], g['solve_B'])
    assert np.abs(X - g['solve_X']).max() < 1e-9  # incl. the singular ones (lstsq branch)
