"""Generate tests/golden/*.npz from the REAL reference (fgnt/pb_bss).

Run in the build container (where /root/reference exists):

    python -m oracle.make_golden

Each fixture stores seeded inputs and the outputs of the unmodified reference
functions (imported through oracle/refshim.py).  The committed fixtures pin
both the NumPy oracle (tests/test_oracle_golden.py, CPU) and the HIP path
(tests/test_gpu_golden.py) on boxes where the reference itself is absent.
"""
import os
import warnings

import numpy as np

from oracle import refshim, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')


def _save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def cacgmm_cases():
    from pb_bss.distribution import CACGMMTrainer
    cases = [
        # name, F, T, D, K, iterations, fit kwargs
        ('cacgmm_f5_t60_d3_k2', 5, 60, 3, 2, 5, {}),
        ('cacgmm_f9_t120_d8_k3', 9, 120, 8, 3, 10, {}),
        ('cacgmm_f6_t80_d6_k3_trace', 6, 80, 6, 3, 4, dict(covariance_norm='trace')),
        ('cacgmm_f6_t80_d4_k2_nonorm', 6, 80, 4, 2, 4, dict(covariance_norm=False)),
        ('cacgmm_f4_t70_d5_k3_uniform', 4, 70, 5, 3, 4, dict(weight_constant_axis=-2)),
        ('cacgmm_f7_t64_d4_k2_shared', 7, 64, 4, 2, 4, dict(weight_constant_axis=(-3,))),
        ('cacgmm_f7_t64_d4_k3_shared_ft', 7, 64, 4, 3, 3, dict(weight_constant_axis=(-3, -1))),
    ]
    for name, F, T, D, K, iters, kw in cases:
        Y, init = synth.make_stft(F, T, D, K, seed=len(name))
        Y128 = Y.astype(np.complex128)  # the float64 path of the reference
        model = CACGMMTrainer().fit(Y128, initialization=init, iterations=iters, **kw)
        aff, q = model.predict(Y128, return_quadratic_form=True)
        _save(name, Y=Y, init=init, iterations=iters,
              weight=model.weight,
              eigvec=model.cacg.covariance_eigenvectors,
              eigval=model.cacg.covariance_eigenvalues,
              covariance=model.cacg.covariance,
              affiliation=aff, quadratic_form=q,
              log_likelihood=model.log_likelihood(Y128),
              kwargs=np.array(repr(kw)))
    # saliency + zero frames + rank-deficient (floor) + batch axis
    Y, init = synth.make_stft(5, 90, 4, 2, seed=3)
    Y[:, 7] = 0
    rng = np.random.default_rng(0)
    sal = rng.uniform(0.2, 1.0, size=(5, 90))
    Y128 = Y.astype(np.complex128)
    model = CACGMMTrainer().fit(Y128, initialization=init, iterations=4, saliency=sal)
    _save('cacgmm_saliency_zero_frame', Y=Y, init=init, iterations=4, saliency=sal,
          weight=model.weight, eigvec=model.cacg.covariance_eigenvectors,
          eigval=model.cacg.covariance_eigenvalues,
          covariance=model.cacg.covariance, affiliation=model.predict(Y128))
    Y, init = synth.make_rank_deficient(4, 80, 5, 2, rank=2, seed=1)
    Y128 = Y.astype(np.complex128)
    model = CACGMMTrainer().fit(Y128, initialization=init, iterations=2)
    _save('cacgmm_rank_deficient', Y=Y, init=init, iterations=2,
          weight=model.weight, eigvec=model.cacg.covariance_eigenvectors,
          eigval=model.cacg.covariance_eigenvalues,
          covariance=model.cacg.covariance, affiliation=model.predict(Y128))
    Yb = np.stack([synth.make_stft(3, 50, 4, 2, seed=s)[0] for s in (1, 2)])
    ib = np.stack([synth.make_stft(3, 50, 4, 2, seed=s)[1] for s in (1, 2)])
    model = CACGMMTrainer().fit(Yb.astype(np.complex128), initialization=ib, iterations=3)
    _save('cacgmm_batch_axis', Y=Yb, init=ib, iterations=3,
          weight=model.weight, eigvec=model.cacg.covariance_eigenvectors,
          eigval=model.cacg.covariance_eigenvalues,
          covariance=model.cacg.covariance,
          affiliation=model.predict(Yb.astype(np.complex128)))


def cacg_cases():
    from pb_bss.distribution.complex_angular_central_gaussian import (
        ComplexAngularCentralGaussianTrainer, normalize_observation)
    from pb_bss.distribution.mixture_model_utils import estimate_mixture_weight
    # doctest KAT of _fit (complex_angular_central_gaussian.py:278-289)
    y = np.array([[1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 1, 0]], dtype=np.complex128).T
    q = np.array([[1, 0], [1, 0], [1, 0], [1, 0]], dtype=np.float64).T
    m = ComplexAngularCentralGaussianTrainer()._fit(y=y, saliency=None, quadratic_form=q)
    _save('cacg_fit_doctest', y=y, quadratic_form=q,
          eigvec=m.covariance_eigenvectors, eigval=m.covariance_eigenvalues,
          covariance=m.covariance)
    # one M-step + log pdf on random data with a class axis
    rng = np.random.default_rng(5)
    Y = (rng.standard_normal((4, 70, 5)) + 1j * rng.standard_normal((4, 70, 5)))
    yn = normalize_observation(Y)
    sal = rng.uniform(size=(4, 3, 70))
    qf = rng.uniform(0.5, 2.0, size=(4, 3, 70))
    m = ComplexAngularCentralGaussianTrainer()._fit(y=yn[:, None], saliency=sal, quadratic_form=qf)
    lp, q2 = m._log_pdf(yn[:, None])
    _save('cacg_m_step_log_pdf', Y=Y, yn=yn, saliency=sal, quadratic_form=qf,
          eigvec=m.covariance_eigenvectors, eigval=m.covariance_eigenvalues,
          covariance=m.covariance, log_pdf=lp, q_out=q2)
    # plain cACG fit (no mixture), test_complex_angular_central_gaussian.py style
    # (the reference's fit only works without independent axes:
    #  `np.ones(*independent, N)`, complex_angular_central_gaussian.py:235)
    m = ComplexAngularCentralGaussianTrainer().fit(Y[0], iterations=5)
    _save('cacg_trainer_fit', Y=Y[0], covariance=m.covariance,
          eigval=m.covariance_eigenvalues, log_pdf=m.log_pdf(Y[0]))
    aff = np.array([[0.4, 1, 0.4], [0.6, 0, 0.6]])
    _save('mixture_weight_doctest', affiliation=aff,
          w_default=estimate_mixture_weight(aff),
          w_axis_m2=estimate_mixture_weight(aff, weight_constant_axis=-2),
          w_stack_m3=estimate_mixture_weight([aff, aff], weight_constant_axis=-3))


def beamformer_cases():
    from pb_bss.extraction import beamformer as bf
    from pb_bss.extraction.beamformer_wrapper import get_bf_vector
    rng = np.random.default_rng(11)
    F, T, D, K = 17, 90, 6, 3

    def cn(*s):
        return rng.standard_normal(s) + 1j * rng.standard_normal(s)

    X = cn(F, D, T)
    mask = rng.uniform(size=(F, K, T))
    psd = bf.get_power_spectral_density_matrix(X, mask)
    psd_nonorm = bf.get_power_spectral_density_matrix(X, mask, normalize=False)
    psd_plain = bf.get_power_spectral_density_matrix(X)
    psd_2d = bf.get_power_spectral_density_matrix(X, mask[:, 0])
    psd_src0 = bf.get_power_spectral_density_matrix(
        X, mask.transpose(1, 0, 2), source_dim=0)
    target = psd[:, 0]
    noise = psd[:, 1] + psd[:, 2]
    out = dict(X=X, mask=mask, psd=psd, psd_nonorm=psd_nonorm, psd_plain=psd_plain,
               psd_2d=psd_2d, psd_src0=psd_src0, target=target, noise=noise)
    out['gev'] = bf._get_gev_vector(target, noise)  # SciPy loop (Cython not built here)
    out['pca'] = bf.get_pca_vector(target)
    w_s, ref = bf.get_mvdr_vector_souden(target, noise, return_ref_channel=True)
    out['mvdr_souden'] = w_s
    out['mvdr_souden_ref'] = ref
    out['mvdr_souden_ch1'] = bf.get_mvdr_vector_souden(target, noise, ref_channel=1)
    atf = bf.get_pca_vector(target)
    out['mvdr'] = bf.get_mvdr_vector(atf, noise)
    out['ban'] = bf.blind_analytic_normalization(out['gev'], noise)
    out['applied'] = bf.apply_beamforming_vector(w_s, X)
    out['wmwf'] = bf.get_wmwf_vector(target, noise)
    out['wmwf_mu3_ch2'] = bf.get_wmwf_vector(target, noise, reference_channel=2, distortion_weight=3.)
    out['wmwf_freqdep'] = bf.get_wmwf_vector(target, noise, reference_channel=0,
                                             distortion_weight='frequency_dependent')
    out['ref_channel_fn'] = bf.get_optimal_reference_channel(
        np.linalg.solve(noise, target), target, noise)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for name in ['pca', 'gev', 'gev+ban', 'mvdr_souden', 'mvdr_souden+ban',
                     'pca+mvdr', 'scaled_gev_atf+mvdr', 'rank1_pca+mvdr_souden',
                     'rank1_gev+mvdr_souden+ban', 'rank1_gev+gev', 'rank1_pca+gev',
                     'wmwf', 'rank1_gev+wmwf+ban', 'ch2']:
            out['bf__' + name.replace('+', '__')] = get_bf_vector(name, target, noise)
    _save('beamformer_f17_d6', **out)
    # MVDR-Souden known answer (tests/test_extraction/test_beamformer.py:185-209)
    obs = np.array([[0, 0, 1], [0, 0.1, 1], [0.1, 0, 1]])
    pxx = obs.T.conj() @ obs
    pnn = np.eye(3)
    w, = bf.get_mvdr_vector_souden(pxx[None], pnn[None])
    from pb_bss.math.solve import stable_solve
    A = cn(5, 4, 4)
    Bm = cn(5, 4, 4)
    A[2, 1, :] = 0
    A[3] = 0
    _save('mvdr_souden_kat', pxx=pxx, pnn=pnn, w=w, w_repr=np.array(repr(w)),
          solve_A=A, solve_B=Bm, solve_X=stable_solve(A, Bm))


def dhtv_cases():
    from pb_bss.permutation_alignment import DHTVPermutationAlignment
    rng = np.random.default_rng(7)
    out = {}
    for tag, size, K, T in [('s512_k2', 512, 2, 60), ('s1024_k3', 1024, 3, 48)]:
        F = size // 2 + 1
        act = rng.uniform(size=(K, T)) ** 4
        mask = act[:, None, :] * rng.uniform(0.5, 1.0, size=(K, F, T)) \
            + 0.05 * rng.uniform(size=(K, F, T))
        mask /= mask.sum(0, keepdims=True)
        perm = np.stack([rng.permutation(K) for _ in range(F)], 1)
        pm = mask[perm, range(F)].astype(np.float32)  # stored compactly; exact upcast in tests
        solver = DHTVPermutationAlignment.from_stft_size(size)
        m64 = pm.astype(np.float64)
        out[tag + '_mask'] = pm
        out[tag + '_mapping'] = solver.calculate_mapping(m64)
        out[tag + '_plan'] = np.asarray(solver.alignment_plan)
        out[tag + '_aligned_sum'] = solver.apply_mapping(m64, out[tag + '_mapping']).sum(-1)
        solver.algorithm = 'optimal'
        out[tag + '_mapping_optimal'] = solver.calculate_mapping(m64)
        for metric in ('multiply', 'euclidean'):
            for alg in ('greedy', 'optimal'):
                sv = DHTVPermutationAlignment.from_stft_size(size, metric)
                sv.algorithm = alg
                out[f'{tag}_mapping_{metric}_{alg}'] = sv.calculate_mapping(m64)
    _save('dhtv_alignment', **out)


def pairwise_alignment_cases():
    """Greedy / Oracle solvers and _mapping_from_score_matrix of the unmodified reference."""
    from pb_bss.permutation_alignment import (GreedyPermutationAlignment,
                                              OraclePermutationAlignment,
                                              _mapping_from_score_matrix, _ScoreMatrix)
    rng = np.random.default_rng(11)
    out = {}
    for tag, K, F, T in [('k2', 2, 33, 40), ('k3', 3, 65, 50), ('k4', 4, 17, 30)]:
        act = rng.uniform(size=(K, T)) ** 4
        ref = act[:, None, :] * rng.uniform(0.5, 1.0, size=(K, F, T)) \
            + 0.05 * rng.uniform(size=(K, F, T))
        ref /= ref.sum(0, keepdims=True)
        perm = np.stack([rng.permutation(K) for _ in range(F)], 1)
        mask = ref[perm, range(F)] * rng.uniform(0.8, 1.2, size=(K, F, T))
        mask = mask.astype(np.float32).astype(np.float64)
        ref = ref.astype(np.float32).astype(np.float64)
        out[tag + '_mask'] = mask.astype(np.float32)
        out[tag + '_reference'] = ref.astype(np.float32)
        for metric in ('cos', 'multiply', 'euclidean'):
            out[f'{tag}_{metric}_scores'] = getattr(_ScoreMatrix, metric)(mask, ref)
            out[f'{tag}_{metric}_greedy'] = GreedyPermutationAlignment(
                similarity_metric=metric).calculate_mapping(mask)
            for alg in ('greedy', 'optimal'):
                out[f'{tag}_{metric}_oracle_{alg}'] = OraclePermutationAlignment(
                    similarity_metric=metric, algorithm=alg).calculate_mapping(mask, ref)
    # block masks in the style of the class doctests (:628-676, :727-775): exact ties
    K, F, T = 3, 25, 6
    blocks = np.zeros((K, F, T))
    for k in range(K):
        blocks[k, 5:, 2 * k:2 * k + 2] = 1
    ref = blocks.copy()
    perm = np.stack([rng.permutation(K) if 10 <= f < 14 else np.arange(K) for f in range(F)], 1)
    perm[:, 14:] = np.array([2, 1, 0])[:, None]
    mask = blocks[perm, range(F)]
    out['blocks_mask'] = mask
    out['blocks_reference'] = ref
    out['blocks_greedy'] = GreedyPermutationAlignment().calculate_mapping(mask)
    out['blocks_oracle'] = OraclePermutationAlignment().calculate_mapping(mask, ref)
    out['blocks_greedy_cos'] = GreedyPermutationAlignment('cos').calculate_mapping(mask)
    out['blocks_oracle_cos_greedy'] = OraclePermutationAlignment('cos', 'greedy') \
        .calculate_mapping(mask, ref)
    sc = rng.normal(size=(40, 4, 4))
    sc[:5] = np.round(sc[:5])  # ties
    out['scores'] = sc
    out['scores_greedy'] = _mapping_from_score_matrix(sc, 'greedy')
    out['scores_optimal'] = _mapping_from_score_matrix(sc, 'optimal')
    _save('pairwise_alignment', **out)


def gmm_cases():
    """GMMTrainer (spherical covariances) of the unmodified reference."""
    from pb_bss.distribution import GMMTrainer
    Y, e, init = synth.make_joint(5, 90, 3, 3, 12, seed=17)
    e64 = (1.5 * e + 0.3).astype(np.float32).astype(np.float64)  # not unit-norm, not centred
    flat = e64.reshape(-1, 12)
    i0 = init.transpose(1, 0, 2).reshape(3, -1)
    m = GMMTrainer().fit(flat, initialization=i0, iterations=8, covariance_type='spherical')
    _save('gmm_n450_e12_k3', y=e64.reshape(-1, 12).astype(np.float32), init=i0, iterations=8,
          mean=m.gaussian.mean, covariance=m.gaussian.covariance, weight=m.weight,
          affiliation=m.predict(flat))
    # NB the reference's SphericalGaussian.log_pdf does not broadcast over independent axes
    # (gaussian.py:110-113 flattens the log-determinant), so GMMTrainer is pinned on flat data
    sal = np.abs(Y[..., 0]).astype(np.float64).reshape(-1)
    fixed = np.array([0.05, 0.2, 0.1])
    out = dict(y=flat.astype(np.float32), init=i0, saliency=sal, iterations=5, fixed=fixed)
    for tag, kw in [('sal', dict(saliency=sal)),
                    ('uniform', dict(weight_constant_axis=-2)),
                    ('fixed', dict(fixed_covariance=fixed))]:
        m = GMMTrainer().fit(flat, initialization=i0, iterations=5,
                             covariance_type='spherical', **kw)
        out.update({f'{tag}_mean': m.gaussian.mean, f'{tag}_covariance': m.gaussian.covariance,
                    f'{tag}_weight': m.weight, f'{tag}_affiliation': m.predict(flat)})
    out['fit_predict'] = GMMTrainer().fit_predict(flat, initialization=i0, iterations=3,
                                                  covariance_type='spherical')
    _save('gmm_variants_n450_e12_k3', **out)
    # full covariances (the default covariance_type)
    m = GMMTrainer().fit(flat, initialization=i0, iterations=6)
    fixed_full = np.stack([np.eye(12) * v + 0.01 for v in (0.05, 0.2, 0.1)])
    mf = GMMTrainer().fit(flat, initialization=i0, iterations=4, saliency=sal,
                          fixed_covariance=fixed_full)
    _save('gmm_full_n450_e12_k3', y=flat.astype(np.float32), init=i0, iterations=6,
          mean=m.gaussian.mean, covariance=m.gaussian.covariance, weight=m.weight,
          affiliation=m.predict(flat), saliency=sal, fixed=fixed_full,
          fixed_mean=mf.gaussian.mean, fixed_weight=mf.weight, fixed_affiliation=mf.predict(flat),
          fit_predict=GMMTrainer().fit_predict(flat, initialization=i0, iterations=3))


def cwmm_cases():
    from pb_bss.distribution.cwmm import CWMMTrainer
    from pb_bss.distribution.complex_watson import ComplexWatson, ComplexWatsonTrainer
    for name, F, T, D, K, iters, kw in [
            ('cwmm_f6_t120_d6_k3', 6, 120, 6, 3, 8, {}),
            ('cwmm_f5_t90_d3_k2', 5, 90, 3, 2, 6, {}),
            ('cwmm_f4_t80_d8_k2_uniform', 4, 80, 8, 2, 5, dict(weight_constant_axis=-2))]:
        Y, init = synth.make_stft(F, T, D, K, seed=len(name) + 3)
        Y128 = Y.astype(np.complex128)
        model = CWMMTrainer().fit(Y128, initialization=init, iterations=iters, **kw)
        yn = Y128 / np.maximum(np.linalg.norm(Y128, axis=-1, keepdims=True), np.finfo(float).tiny)
        _save(name, Y=Y, init=init, iterations=iters, kwargs=np.array(repr(kw)),
              weight=model.weight, mode=model.complex_watson.mode,
              concentration=model.complex_watson.concentration,
              affiliation=model.predict(Y128),
              log_pdf=model.complex_watson.log_pdf(yn[..., None, :, :]))
    t = ComplexWatsonTrainer(5)
    ev = np.array([0, 1 / 5, 1 / 5 + 1e-4, 0.3, 0.7, 0.9599999, 1])
    ks = np.array([0.0, 1e-3, 0.5, 5.0, 50.0, 300.0, 500.0])
    _save('watson_scalar_functions', eigenvalues=ev,
          ratio_inverse_d5=t.hypergeometric_ratio_inverse(ev),  # doctest complex_watson.py:268-271
          concentrations=ks,
          log_norm_d5=ComplexWatson.log_norm_1f1(ks, 5),
          log_norm_d8=ComplexWatson.log_norm_1f1(ks, 8),
          ratio_d5=t.hypergeometric_ratio(ks))


def embed_cases():
    """N2 (vMF mixture) and N3 (joint spatial+spectral mixtures) fixtures."""
    from pb_bss.distribution import (VMFMMTrainer, GCACGMMTrainer, VMFCACGMMTrainer,
                                     GaussianTrainer, VonMisesFisherTrainer)
    from pb_bss.distribution.von_mises_fisher import VonMisesFisher
    Y, e, init = synth.make_joint(6, 80, 4, 3, 10, seed=3)
    e64 = e.astype(np.float64)
    Y128 = Y.astype(np.complex128)
    flat = e64.reshape(-1, 10)
    i0 = init.transpose(1, 0, 2).reshape(3, -1)
    m = VMFMMTrainer().fit(flat, initialization=i0, iterations=7)
    _save('embed_vmfmm_n480_e10_k3', y=e.reshape(-1, 10), init=i0, iterations=7,
          mean=m.vmf.mean, concentration=m.vmf.concentration, weight=m.weight,
          affiliation=m.predict(flat), log_pdf=m.vmf.log_pdf(flat[None]))
    sal = np.abs(Y128[..., 0])
    m = VMFMMTrainer().fit(e64, initialization=init, iterations=5, saliency=sal,
                           max_concentration=40.)
    _save('embed_vmfmm_indep_f6_t80_e10_k3', y=e, init=init, saliency=sal, iterations=5,
          max_concentration=40., mean=m.vmf.mean, concentration=m.vmf.concentration,
          weight=m.weight, affiliation=m.predict(e64))
    ks = np.array([1e-10, 1e-3, 0.7, 5.0, 37.5, 120.0, 500.0])
    _save('embed_vmf_log_norm', concentrations=ks,
          **{f'log_norm_d{d}': VonMisesFisher(np.ones(d) / np.sqrt(d), ks).log_norm()
             for d in (2, 3, 10, 40)})
    rng = np.random.default_rng(21)
    w = rng.uniform(size=(3, 480))
    g = {ct: GaussianTrainer()._fit(flat[None], saliency=w, covariance_type=ct)
         for ct in ('spherical', 'diagonal', 'full')}
    v = VonMisesFisherTrainer()._fit(flat[None], saliency=w, min_concentration=1e-10,
                                     max_concentration=500)
    _save('embed_single_fits', y=e.reshape(-1, 10), saliency=w,
          vmf_mean=v.mean, vmf_concentration=v.concentration,
          **{f'{ct}_{k}': getattr(g[ct], k) for ct in g for k in ('mean', 'covariance')},
          **{f'{ct}_log_pdf': g[ct].log_pdf(flat[None]) for ct in g})
    for name, trainer, kind, emb, kw in [
            ('embed_gcacgmm_spherical', GCACGMMTrainer, 'gaussian', e, {}),
            ('embed_gcacgmm_weights', GCACGMMTrainer, 'gaussian', e,
             dict(spatial_weight=0.7, spectral_weight=1.3, weight_constant_axis=(-3,))),
            ('embed_gcacgmm_inline_pa', GCACGMMTrainer, 'gaussian', e,
             dict(inline_permutation_alignment=True, weight_constant_axis=(-3, -1))),
            ('embed_vmfcacgmm', VMFCACGMMTrainer, 'vmf', (2 * e).astype(np.float32), {})]:
        emb64 = emb.astype(np.float64)
        m = trainer().fit(Y128, emb64, initialization=init, iterations=5, **kw)
        spec = m.gaussian if kind == 'gaussian' else m.vmf
        extra = (dict(covariance=spec.covariance) if kind == 'gaussian'
                 else dict(concentration=spec.concentration))
        _save(name, Y=Y, embedding=emb, init=init, iterations=5, kwargs=np.array(repr(kw)),
              weight=np.asarray(m.weight), mean=spec.mean,
              eigvec=m.cacg.covariance_eigenvectors, eigval=m.cacg.covariance_eigenvalues,
              affiliation=m.predict(Y128, emb64), **extra)


def beamformer_extra_cases():
    """N4: the rest of extraction/beamformer.py."""
    from pb_bss.extraction import beamformer as bf
    rng = np.random.default_rng(31)
    F, D, K, T = 19, 5, 2, 40

    def cn(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)

    def psd(n):
        a = cn(n, D, 2 * D)
        return a @ a.conj().swapaxes(-1, -2) / (2 * D)

    target, noise = psd(F), psd(F) + 0.1 * np.eye(D)
    atf = cn(K, F, D)
    w = cn(F, D)
    wb = cn(3, F, D)
    vt = cn(T, F, D)
    mix = cn(F, D, T).astype(np.complex64)
    noise_sing = noise.copy()
    noise_sing[4] = np.outer(atf[0, 4], atf[0, 4].conj())     # rank one: lstsq branch
    pca_all = bf.get_pca(target, return_all_vecs=True)
    pca_one = bf.get_pca(target)
    _save('beamformer_extra_f19_d5',
          target=target, noise=noise, atf=atf, w=w, wb=wb, vt=vt, mix=mix, noise_sing=noise_sing,
          lcmv_10=bf.get_lcmv_vector(atf, [1, 0], noise),
          lcmv_01=bf.get_lcmv_vector(atf, [0, 1], noise),
          lcmv_sing=bf.get_lcmv_vector(atf, [1, 0], noise_sing),
          merl=bf.get_mvdr_vector_merl(target, noise),
          distortionless=bf.distortionless_normalization(w, atf[0], noise),
          postfilter=bf.mvdr_snr_postfilter(w, target, noise),
          zero_degree=bf.zero_degree_normalization(wb, 2),
          phase_2d=bf.phase_correction(w), phase_3d=bf.phase_correction(wb),
          conditioned=bf.condition_covariance(target, 0.05),
          online=bf.apply_online_beamforming_vector(vt, mix),
          pca_all_vec=pca_all[0], pca_all_val=pca_all[1], pca_vec=pca_one[0], pca_val=pca_one[1])


def sampler_cases():
    """Seeded draws of the reference's host-side samplers (global NumPy RNG)."""
    from pb_bss.distribution.cacgmm import sample_cacgmm
    from pb_bss.distribution import ComplexAngularCentralGaussian
    rng = np.random.default_rng(4)
    a = rng.standard_normal((2, 3, 6)) + 1j * rng.standard_normal((2, 3, 6))
    cov = a @ a.conj().swapaxes(-1, -2)
    weight = np.array([0.3, 0.7])
    np.random.seed(11)
    x, labels = sample_cacgmm(50, weight, cov, return_label=True)
    np.random.seed(12)
    m = ComplexAngularCentralGaussian.from_covariance(cov[0].copy())
    y = m.sample(size=(20,))
    _save('sampler_draws', covariance=cov, weight=weight, x=x, labels=labels,
          eigvec=m.covariance_eigenvectors, eigval=m.covariance_eigenvalues, y=y)


def joint_covariance_cases():
    """GCACGMMTrainer(covariance_type='full' | 'diagonal') (gcacgmm.py:141, :297-303 ->
    GaussianTrainer._fit, gaussian.py:152-193): same layout as the 'embed_gcacgmm_*' fixtures."""
    from pb_bss.distribution import GCACGMMTrainer
    Y, e, init = synth.make_joint(6, 80, 4, 3, 10, seed=3)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    rng = np.random.default_rng(77)
    A = rng.standard_normal((3, 10, 10))
    fixed_full = A @ A.swapaxes(-1, -2) / 10 + np.eye(10)
    for name, kw in [
            ('embed_gcacgmm_full', dict(covariance_type='full')),
            ('embed_gcacgmm_diagonal', dict(covariance_type='diagonal')),
            ('embed_gcacgmm_full_weights', dict(covariance_type='full', spatial_weight=0.7,
                                                spectral_weight=1.3, weight_constant_axis=(-3,))),
            ('embed_gcacgmm_diagonal_pa', dict(covariance_type='diagonal',
                                               inline_permutation_alignment=True,
                                               weight_constant_axis=(-3, -1))),
            ('embed_gcacgmm_full_fixed', dict(covariance_type='full',
                                              fixed_covariance=fixed_full))]:
        m = GCACGMMTrainer().fit(Y128, e64, initialization=init, iterations=5, **kw)
        kw_repr = {k: v for k, v in kw.items() if k != 'fixed_covariance'}
        extra = {}
        if 'fixed_covariance' in kw:
            extra['fixed_covariance'] = kw['fixed_covariance']
        _save(name, Y=Y, embedding=e, init=init, iterations=5, kwargs=np.array(repr(kw_repr)),
              weight=np.asarray(m.weight), mean=m.gaussian.mean, covariance=m.gaussian.covariance,
              eigvec=m.cacg.covariance_eigenvectors, eigval=m.cacg.covariance_eigenvalues,
              affiliation=m.predict(Y128, e64), **extra)


def gev_eig_cases():
    """`get_gev_vector(..., use_eig=True)`: the zggev module compiled from the reference's own
    c_eig.pyx (oracle/refshim.py:build_cython) AND the scipy.linalg.eig fallback loop, on
    Hermitian-definite pencils, on a Hermitian pencil whose noise matrix is indefinite, and on
    pencils that are not Hermitian at all."""
    import pb_bss.extraction.beamformer as bf
    assert bf.c_eig_available and bf.c_gev_available, 'build the .pyx first (refshim.load_cython)'
    rng = np.random.default_rng(23)

    def cn(*s):
        return rng.standard_normal(s) + 1j * rng.standard_normal(s)

    out = {}
    for tag, F, D in (('hpd_d6', 17, 6), ('hpd_d8', 9, 8), ('hpd_d3', 5, 3)):
        X, Nn = cn(F, D, 4 * D), cn(F, D, 5 * D)
        out[tag + '_target'] = X @ X.conj().swapaxes(-1, -2)
        out[tag + '_noise'] = Nn @ Nn.conj().swapaxes(-1, -2)
    X = cn(7, 5, 12)
    H = cn(7, 5, 5)
    out['indef_d5_target'] = X @ X.conj().swapaxes(-1, -2)
    out['indef_d5_noise'] = H + H.conj().swapaxes(-1, -2)          # Hermitian, indefinite
    out['general_d6_target'] = cn(11, 6, 6)                          # not Hermitian
    out['general_d6_noise'] = cn(11, 6, 6) + 3 * np.eye(6)
    out['general_d2_target'] = cn(4, 2, 2)
    out['general_d2_noise'] = cn(4, 2, 2) + 2 * np.eye(2)
    for tag in sorted({k.rsplit('_', 1)[0] for k in out}):
        t, n = out[tag + '_target'], out[tag + '_noise']
        out[tag + '_w_cython'] = bf.get_gev_vector(t, n, use_eig=True)        # c_eig.pyx / zggev
        out[tag + '_w_scipy'] = bf._get_gev_vector(t, n, use_eig=True)       # scipy.linalg.eig loop
        vals, _ = bf._cythonized_eig(t, n)
        out[tag + '_lambda'] = vals[np.arange(t.shape[0]), np.argmax(vals, axis=1)]
    # the Hermitian-definite solver of the compiled get_gev_vector.pyx (zhegvd) on the same pencil
    out['hpd_d6_w_zhegvd'] = bf.get_gev_vector(out['hpd_d6_target'], out['hpd_d6_noise'])
    _save('gev_use_eig', **out)


def cacgmm_single_precision_cases():
    """The reference's OWN single-precision path: a complex64 observation with an ndarray
    initialisation runs the whole EM in complex64 / float32 (cacgmm.py:226-227).  These fixtures
    pin the packed-FP32 kernel (pbbss_em_opts.precision = F32) per step -- EM trajectories are
    chaotic, single precision cannot be compared over many iterations."""
    from pb_bss.distribution import CACGMMTrainer
    out = {}
    for tag, F, T, D, K in (('a', 12, 300, 8, 3), ('b', 6, 140, 4, 2), ('c', 5, 500, 6, 4)):
        Y, init = synth.make_stft(F, T, D, K, seed=40 + F)
        assert Y.dtype == np.complex64
        for iters in (1, 2):
            model = CACGMMTrainer().fit(Y, initialization=init, iterations=iters)
            aff = model.predict(Y)
            assert aff.dtype == np.float32 and model.cacg.covariance_eigenvalues.dtype == np.float32
            m64 = CACGMMTrainer().fit(Y.astype(np.complex128), initialization=init,
                                      iterations=iters)
            out[f'{tag}_aff32_it{iters}'] = aff
            out[f'{tag}_cov32_it{iters}'] = model.cacg.covariance
            out[f'{tag}_weight32_it{iters}'] = model.weight
            out[f'{tag}_aff64_it{iters}'] = m64.predict(Y.astype(np.complex128))
        out[f'{tag}_Y'] = Y
        out[f'{tag}_init'] = init
    _save('cacgmm_single_precision_path', **out)


def cacgmm_single_precision_mask_case():
    """source_activity_mask on the reference's single-precision path (cacgmm.py:269-271 ->
    mixture_model_utils.py:39-41): pins the mask handling of the packed-FP32 kernel, per step."""
    from pb_bss.distribution import CACGMMTrainer
    out = {}
    F, T, D, K = 6, 200, 5, 3
    Y, init = synth.make_stft(F, T, D, K, seed=77)
    rng = np.random.default_rng(78)
    mask = rng.uniform(size=(F, K, T)) > 0.3
    mask[:, 0, :] |= ~mask.any(axis=1)          # every frame keeps at least one active class
    mask[2, :, 50:60] = False                   # ... except a stretch of bin 2: all classes off
    assert Y.dtype == np.complex64
    for iters in (1, 2):
        model = CACGMMTrainer().fit(Y, initialization=init, iterations=iters,
                                    source_activity_mask=mask)
        aff = model.predict(Y)
        assert aff.dtype == np.float32
        m64 = CACGMMTrainer().fit(Y.astype(np.complex128), initialization=init, iterations=iters,
                                  source_activity_mask=mask)
        out[f'aff32_it{iters}'] = aff
        out[f'weight32_it{iters}'] = model.weight
        out[f'aff64_it{iters}'] = m64.predict(Y.astype(np.complex128))
    out['Y'], out['init'], out['mask'] = Y, init, mask
    _save('cacgmm_single_precision_mask', **out)


def cacgmm_single_precision_many_classes():
    """Five and six classes on the reference's single-precision path (cacgmm.py:226-227): pins the
    K > 4 instantiations of the packed-FP32 kernel, per step."""
    from pb_bss.distribution import CACGMMTrainer
    out = {}
    for tag, F, T, D, K in (('k5', 5, 220, 6, 5), ('k6', 4, 300, 8, 6)):
        Y, init = synth.make_stft(F, T, D, K, seed=90 + K)
        assert Y.dtype == np.complex64
        for iters in (1, 2):
            model = CACGMMTrainer().fit(Y, initialization=init, iterations=iters)
            aff = model.predict(Y)
            assert aff.dtype == np.float32
            m64 = CACGMMTrainer().fit(Y.astype(np.complex128), initialization=init,
                                      iterations=iters)
            out[f'{tag}_aff32_it{iters}'] = aff
            out[f'{tag}_weight32_it{iters}'] = model.weight
            out[f'{tag}_aff64_it{iters}'] = m64.predict(Y.astype(np.complex128))
        out[f'{tag}_Y'] = Y
        out[f'{tag}_init'] = init
    _save('cacgmm_single_precision_k56', **out)


def public_names_cases():
    """Round 6: the remaining public names of the hot-path modules -- rank-one estimates
    (beamformer_wrapper.py:11-69), get_single_source_bf_vector (extraction/__init__.py:4), the
    stand-alone inline-PA posterior (mixture_model_utils.py:58-130), log_pdf_to_affiliation and
    estimate_mixture_weight called directly (mixture_model_utils.py:7-55, :133-203)."""
    import pb_bss.extraction as ex
    from pb_bss.extraction import beamformer_wrapper as bw
    from pb_bss.distribution import mixture_model_utils as mmu
    rng = np.random.default_rng(61)
    F, D, K, T = 11, 4, 3, 50

    def cn(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)

    def psd(n):
        a = cn(n, D, 2 * D)
        return a @ a.conj().swapaxes(-1, -2) / (2 * D)

    target, noise = psd(F), psd(F) + 0.1 * np.eye(D)
    spatial = 3.0 * rng.standard_normal((F, K, T))
    spectral = 3.0 * rng.standard_normal((F, K, T))
    # make the search matter: per bin a hidden permutation of a common pattern plus noise
    base = 4.0 * rng.standard_normal((K, T))
    for f in range(F):
        spatial[f] += base[rng.permutation(K)]
        spectral[f] += base
    weight_k = rng.uniform(0.2, 1.0, size=(K, 1))
    weight_k /= weight_k.sum()
    weight_fk = rng.uniform(0.2, 1.0, size=(F, K, 1))
    weight_fk /= weight_fk.sum(1, keepdims=True)
    activity = rng.uniform(size=(F, K, T)) > 0.2
    activity[:, 0] = True
    aff = rng.uniform(size=(2, F, K, T))
    aff /= aff.sum(-2, keepdims=True)
    sal = rng.uniform(0.1, 1.0, size=(2, F, T))
    _save('public_names_r06',
          target=target, noise=noise, spatial=spatial, spectral=spectral, weight_k=weight_k,
          weight_fk=weight_fk, activity=activity, aff=aff, sal=sal,
          pca_rank1=bw.get_pca_rank_one_estimate(target),
          gev_rank1=bw.get_gev_rank_one_estimate(target, noise),
          single_source_gev_ban=ex.get_single_source_bf_vector('gev+ban', target, noise),
          single_source_rank1=ex.get_single_source_bf_vector('rank1_gev+mvdr_souden', target, noise),
          inline_pa_k=mmu.log_pdf_to_affiliation_for_integration_models_with_inline_pa(
              weight_k, spatial, spectral),
          inline_pa_fk_eps=mmu.log_pdf_to_affiliation_for_integration_models_with_inline_pa(
              weight_fk, spatial, spectral, affiliation_eps=1e-3),
          inline_pa_act=mmu.log_pdf_to_affiliation_for_integration_models_with_inline_pa(
              weight_k, spatial, spectral, source_activity_mask=activity),
          l2a_k=mmu.log_pdf_to_affiliation(weight_k, spatial),
          l2a_fk_act_eps=mmu.log_pdf_to_affiliation(weight_fk, spatial, source_activity_mask=activity,
                                                    affiliation_eps=1e-4),
          l2a_batched=mmu.log_pdf_to_affiliation(weight_k, np.stack([spatial, spectral])),
          mixw_n=mmu.estimate_mixture_weight(aff, weight_constant_axis=-1),
          mixw_fn_sal=mmu.estimate_mixture_weight(aff, saliency=sal, weight_constant_axis=(-3, -1)),
          mixw_f=mmu.estimate_mixture_weight(aff, weight_constant_axis=(-3,)),
          mixw_class=mmu.estimate_mixture_weight(aff, weight_constant_axis=-2),
          mixw_outer=mmu.estimate_mixture_weight(aff, weight_constant_axis=(0, -1)))


def main():
    """python -m oracle.make_golden            -> every fixture of the pure-Python reference
    python -m oracle.make_golden f32k56     -> tests/golden/cacgmm_single_precision_k56.npz
    python -m oracle.make_golden f32mask    -> tests/golden/cacgmm_single_precision_mask.npz
    python -m oracle.make_golden f32        -> tests/golden/cacgmm_single_precision_path.npz
    python -m oracle.make_golden joint_cov  -> tests/golden/embed_gcacgmm_{full,diagonal}*.npz
    python -m oracle.make_golden public_names -> tests/golden/public_names_r06.npz
    python -m oracle.make_golden gev_eig    -> tests/golden/gev_use_eig.npz only (own process:
    the reference's Cython modules must be injected BEFORE pb_bss.extraction.beamformer is
    imported, and the other fixtures are defined as the Cython-less reference's output)."""
    import sys
    os.makedirs(OUT, exist_ok=True)
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    if sys.argv[1:] == ['joint_cov']:
        refshim.load()
        joint_covariance_cases()
        return
    if sys.argv[1:] == ['f32']:
        refshim.load()
        cacgmm_single_precision_cases()
        return
    if sys.argv[1:] == ['f32k56']:
        refshim.load()
        cacgmm_single_precision_many_classes()
        return
    if sys.argv[1:] == ['f32mask']:
        refshim.load()
        cacgmm_single_precision_mask_case()
        return
    if sys.argv[1:] == ['public_names']:
        refshim.load()
        public_names_cases()
        return
    if sys.argv[1:] == ['gev_eig']:
        refshim.load_cython()
        gev_eig_cases()
        return
    refshim.load()
    cacgmm_cases()
    cacgmm_single_precision_cases()
    cacg_cases()
    beamformer_cases()
    dhtv_cases()
    pairwise_alignment_cases()
    gmm_cases()
    cwmm_cases()
    embed_cases()
    beamformer_extra_cases()
    sampler_cases()
    public_names_cases()


if __name__ == '__main__':
    main()
