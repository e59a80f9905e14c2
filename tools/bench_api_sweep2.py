#!/usr/bin/env python3
"""Second part of the API sweep (tools/bench_api_sweep.py): joint models and their options, the
weight modes of the Watson / vMF / Gaussian mixtures, saliencies, a batch of utterances, the
single-distribution trainers and pipeline.separate.  PBBSS_SWEEP=2 selects it in
tools/prof_api_sweep.sh."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd import _lib, pipeline
from pb_bss_amd.testing import synth
from pb_bss_amd.distribution import (CACGMMTrainer, CWMMTrainer, VMFMMTrainer, GMMTrainer,
                                     GCACGMMTrainer, VMFCACGMMTrainer, GaussianTrainer,
                                     VonMisesFisherTrainer, ComplexWatsonTrainer,
                                     normalize_observation)


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
        print(f'{name}: FAILED {type(e).__name__}: {str(e)[:200]}')


F, T, D, K, E = 513, 500, 8, 3, 40
rng = np.random.default_rng(0)
Yj, emb, initj = synth.make_joint(F, T, D, K, E, seed=2)
y, e, g0 = _lib.to_device(Yj), _lib.to_device(emb), _lib.to_device(initj)
sal = _lib.to_device(rng.uniform(0.2, 1.0, size=(F, T)))
it = 10
for ct in ('spherical', 'diagonal', 'full'):
    timed(f'GCACGMM fit 10 it, covariance_type={ct}',
          lambda: GCACGMMTrainer().fit(y, e, initialization=g0, iterations=it, covariance_type=ct))
for ax in ((-3,), (-3, -1), -2, (-3, -2, -1)):
    timed(f'GCACGMM fit 10 it, weight_constant_axis={ax}',
          lambda: GCACGMMTrainer().fit(y, e, initialization=g0, iterations=it, weight_constant_axis=ax))
timed('GCACGMM fit 10 it, saliency', lambda: GCACGMMTrainer().fit(y, e, initialization=g0, iterations=it, saliency=sal))
timed('GCACGMM fit 10 it, inline alignment, (-3,)',
      lambda: GCACGMMTrainer().fit(y, e, initialization=g0, iterations=it, weight_constant_axis=(-3,),
                                   inline_permutation_alignment=True))
timed('GCACGMM fit 10 it, spatial_weight=0.5',
      lambda: GCACGMMTrainer().fit(y, e, initialization=g0, iterations=it, spatial_weight=0.5))
mj = GCACGMMTrainer().fit(y, e, initialization=g0, iterations=it)
timed('GCACGMM predict', lambda: mj.predict(y, e))
timed('VMFCACGMM fit 10 it', lambda: VMFCACGMMTrainer().fit(y, e, initialization=g0, iterations=it))
timed('VMFCACGMM fit 10 it, (-3, -1)',
      lambda: VMFCACGMMTrainer().fit(y, e, initialization=g0, iterations=it, weight_constant_axis=(-3, -1)))

for ax in ((-3,), (-3, -1), -2):
    timed(f'CWMM fit 10 it, weight_constant_axis={ax}',
          lambda: CWMMTrainer().fit(y, initialization=g0, iterations=it, weight_constant_axis=ax))
timed('CWMM fit 10 it, saliency', lambda: CWMMTrainer().fit(y, initialization=g0, iterations=it, saliency=sal))
mw = CWMMTrainer().fit(y, initialization=g0, iterations=it)
timed('CWMM predict', lambda: mw.predict(y))

ev = e.reshape(F * T, E)
ge = _lib.to_device(np.ascontiguousarray(np.moveaxis(initj, 1, 0).reshape(K, F * T)))
sn = _lib.to_device(rng.uniform(0.2, 1.0, size=(F * T,)))
timed('VMFMM fit 10 it, saliency', lambda: VMFMMTrainer().fit(ev, initialization=ge, iterations=it, saliency=sn))
timed('VMFMM fit 10 it, weight_constant_axis=-2',
      lambda: VMFMMTrainer().fit(ev, initialization=ge, iterations=it, weight_constant_axis=-2))
mv = VMFMMTrainer().fit(ev, initialization=ge, iterations=it)
timed('VMFMM predict', lambda: mv.predict(ev))
for ct in ('spherical', 'diagonal', 'full'):
    timed(f'GMM fit 10 it, {ct}, saliency',
          lambda: GMMTrainer().fit(ev, initialization=ge, iterations=it, covariance_type=ct, saliency=sn))
    mg = GMMTrainer().fit(ev, initialization=ge, iterations=2, covariance_type=ct)
    timed(f'GMM predict, {ct}', lambda: mg.predict(ev))
    timed(f'GaussianTrainer fit, {ct}', lambda: GaussianTrainer().fit(ev, saliency=sn, covariance_type=ct))
timed('VonMisesFisherTrainer fit', lambda: VonMisesFisherTrainer().fit(ev, saliency=sn))
yn = normalize_observation(y)                     # (F, D, T)
timed('ComplexWatsonTrainer fit', lambda: ComplexWatsonTrainer().fit(yn.permute(0, 2, 1).contiguous(), saliency=sal))

B = 4
Yb = np.stack([synth.make_stft(F, T, D, K, seed=s)[0] for s in range(B)])
gb = np.stack([synth.make_stft(F, T, D, K, seed=s)[1] for s in range(B)])
yb, g0b = _lib.to_device(Yb), _lib.to_device(gb)
timed('CACGMM fit 10 it, batch of 4', lambda: CACGMMTrainer().fit(yb, initialization=g0b, iterations=it))
timed('CACGMM fit_predict 10 it, batch of 4', lambda: CACGMMTrainer().fit_predict(yb, initialization=g0b, iterations=it))
for bf in pipeline.BEAMFORMERS:
    timed(f"pipeline.separate 10 it, '{bf}'", lambda: pipeline.separate(y, g0, iterations=it, stft_size=1024, beamformer=bf))
timed("pipeline.separate 10 it, batch of 4", lambda: pipeline.separate(yb, g0b, iterations=it, stft_size=1024))
