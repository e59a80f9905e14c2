#!/usr/bin/env python3
"""One call of (nearly) every public entry point at utterance size (F = 513, T = 500, D = 8,
K = 3), device tensors in and out, median wall time per call -- meant to be run under
`rocprofv3 --kernel-trace --stats` (tools/prof_api_sweep.sh) to find helper kernels whose time is
out of proportion to their bytes (serial chains, tiny grids)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd import _lib
from pb_bss_amd.testing import synth
from pb_bss_amd import extraction as ex, transform
from pb_bss_amd.distribution import (CACGMMTrainer, CWMMTrainer, VMFMMTrainer, GMMTrainer,
                                     ComplexAngularCentralGaussianTrainer, normalize_observation)
from pb_bss_amd.permutation_alignment import (DHTVPermutationAlignment, GreedyPermutationAlignment,
                                              OraclePermutationAlignment)


def timed(name, fn, reps=5):
    try:
        for _ in range(2):
            fn()
        gc.collect()
        ms = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        print(f'{name}: median {np.median(ms):.3f} ms per call')
    except Exception as e:  # keep sweeping
        print(f'{name}: FAILED {type(e).__name__}: {e}')


F, T, D, K = 513, 500, 8, 3
rng = np.random.default_rng(0)
Y, init = synth.make_stft(F, T, D, K, seed=0)
y, g0 = _lib.to_device(Y), _lib.to_device(init)
sal = _lib.to_device(rng.uniform(0.2, 1.0, size=(F, T)))
act = _lib.to_device(rng.uniform(size=(F, K, T)) > 0.1)
it = 10

timed('normalize_observation', lambda: normalize_observation(y))
timed('CACGMM fit 10 it', lambda: CACGMMTrainer().fit(y, initialization=g0, iterations=it))
timed('CACGMM fit 10 it, saliency', lambda: CACGMMTrainer().fit(y, initialization=g0, iterations=it, saliency=sal))
timed('CACGMM fit 10 it, source_activity_mask',
      lambda: CACGMMTrainer().fit(y, initialization=g0, iterations=it, source_activity_mask=act))
for ax in ((-3,), (-3, -1), -2, (-1,)):
    timed(f'CACGMM fit 10 it, weight_constant_axis={ax}',
          lambda: CACGMMTrainer().fit(y, initialization=g0, iterations=it, weight_constant_axis=ax))
for cn in ('trace', False):
    timed(f'CACGMM fit 10 it, covariance_norm={cn}',
          lambda: CACGMMTrainer().fit(y, initialization=g0, iterations=it, covariance_norm=cn))
timed('CACGMM fit 10 it, weight_constant_axis=(-3,), inline DHTV aligner',
      lambda: CACGMMTrainer().fit(y, initialization=g0, iterations=it, weight_constant_axis=(-3,),
                                  inline_permutation_aligner=DHTVPermutationAlignment.from_stft_size(1024)))
timed('CACGMM fit num_classes=3, 10 it', lambda: CACGMMTrainer().fit(y, num_classes=K, iterations=it))
model = CACGMMTrainer().fit(y, initialization=g0, iterations=it)
timed('CACGMM predict', lambda: model.predict(y))
timed('CACGMM fit from model, 10 it', lambda: CACGMMTrainer().fit(y, initialization=model, iterations=it))
yn = normalize_observation(y)
timed('cACG trainer _fit (one M-step)',
      lambda: ComplexAngularCentralGaussianTrainer().fit(y[:, None], saliency=g0, quadratic_form=None)
      if False else ComplexAngularCentralGaussianTrainer()._fit(yn[:, None], saliency=g0,
                                                                quadratic_form=torch.ones_like(g0)))
timed('CWMM fit 10 it', lambda: CWMMTrainer().fit(y, initialization=g0, iterations=it))

masks = model.predict(y)                          # (F, K, T)
mk = masks.permute(1, 0, 2).contiguous()          # (K, F, T)
timed('DHTV calculate_mapping', lambda: DHTVPermutationAlignment.from_stft_size(1024).calculate_mapping(mk))
timed('DHTV __call__ (align)', lambda: DHTVPermutationAlignment.from_stft_size(1024)(mk))
timed('Greedy __call__', lambda: GreedyPermutationAlignment()(mk))
timed('Oracle __call__', lambda: OraclePermutationAlignment()(mk, mk))

obs = y.permute(0, 2, 1).contiguous()             # (F, D, T)
timed('psd (F,K,T) mask', lambda: ex.get_power_spectral_density_matrix(obs, masks))
timed('psd no mask', lambda: ex.get_power_spectral_density_matrix(obs))
psd = ex.get_power_spectral_density_matrix(obs, masks)   # (F, K, D, D)
tgt = psd[:, 0].contiguous()
noi = (psd[:, 1] + psd[:, 2]).contiguous()
timed('get_gev_vector', lambda: ex.get_gev_vector(tgt, noi))
timed('get_pca_vector', lambda: ex.get_pca_vector(tgt))
timed('get_pca (all)', lambda: ex.get_pca(tgt, return_all_vecs=True))
timed('get_mvdr_vector_souden', lambda: ex.get_mvdr_vector_souden(tgt, noi))
timed('get_mvdr_vector_souden ref_channel=None', lambda: ex.get_mvdr_vector_souden(tgt, noi, ref_channel=None))
atf = ex.get_pca_vector(tgt)
timed('get_mvdr_vector', lambda: ex.get_mvdr_vector(atf, noi))
timed('get_mvdr_vector_merl', lambda: ex.get_mvdr_vector_merl(tgt, noi))
timed('get_wmwf_vector', lambda: ex.get_wmwf_vector(tgt, noi))
timed('get_optimal_reference_channel', lambda: ex.get_optimal_reference_channel(
    ex.get_mvdr_vector_souden(tgt, noi, return_ref_channel=False) if False else tgt.new_ones((F, D, D)), tgt, noi))
w = ex.get_gev_vector(tgt, noi)
timed('blind_analytic_normalization', lambda: ex.blind_analytic_normalization(w, noi))
timed('distortionless_normalization', lambda: ex.distortionless_normalization(w, atf, noi))
timed('mvdr_snr_postfilter', lambda: ex.mvdr_snr_postfilter(w, tgt, noi))
timed('zero_degree_normalization', lambda: ex.zero_degree_normalization(w, 0))
timed('phase_correction', lambda: ex.phase_correction(w))
timed('condition_covariance', lambda: ex.condition_covariance(noi, 1e-3))
timed('apply_beamforming_vector', lambda: ex.apply_beamforming_vector(w, obs))
for name in ('gev+ban', 'mvdr_souden', 'pca', 'rank1_gev+mvdr_souden', 'scaled_gev_atf+mvdr', 'wmwf'):
    timed(f"get_bf_vector('{name}')", lambda: ex.get_bf_vector(name, tgt, noi))
atfs = torch.stack([ex.get_pca_vector(psd[:, k].contiguous()) for k in range(K)], 0)    # (K, F, D)
resp = tgt.new_zeros(K); resp[0] = 1
timed('get_lcmv_vector', lambda: ex.get_lcmv_vector(atfs, resp, noi))
vt = w[None].expand(T, F, D).contiguous()
timed('apply_online_beamforming_vector', lambda: ex.apply_online_beamforming_vector(vt, obs))

sig = _lib.to_device(rng.standard_normal((D, 16000 * 8)))
timed('stft 8 ch x 8 s', lambda: transform.stft(sig, 1024, 256))
S = transform.stft(sig, 1024, 256)
timed('istft 8 ch x 8 s', lambda: transform.istft(S, 1024, 256))

e = _lib.to_device(rng.standard_normal((F * T, 40)).astype(np.float32))
ge = _lib.to_device(np.ascontiguousarray(np.moveaxis(init, 1, 0).reshape(K, F * T)))
timed('VMFMM fit 10 it', lambda: VMFMMTrainer().fit(e, initialization=ge, iterations=it))
for ct in ('spherical', 'diagonal', 'full'):
    timed(f'GMM fit 10 it, {ct}', lambda: GMMTrainer().fit(e, initialization=ge, iterations=it, covariance_type=ct))
