#!/usr/bin/env python3
"""Exercise the kernels that are new in round 3 at BASELINE shapes, for a rocprofv3 kernel trace
(tools/prof_round3_kernels.sh): DHTV frame-slice path (T = 500 and 1000), Watson mixture with
split groups (configs[3]) and at generic sizes (D = 12, K = 8), joint model at D = 12, generic
GEV / BAN chain at D = 29, generic-size EM with the remainder-bin side chain."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing import synth
from pb_bss_amd.distribution import CWMMTrainer, GCACGMMTrainer, CACGMMTrainer
from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
from pb_bss_amd import extraction as ex


def timed(name, fn, reps=5):
    for _ in range(3):   # code-object load and the plan cache are first-call costs
        fn()
    ms = []
    gc.collect()   # a generation-2 collection inside a timed call costs tens of ms with torch imported
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    print(f'{name}: median {np.median(ms):.3f} ms per call (min {min(ms):.3f}, max {max(ms):.3f})')


rng = np.random.default_rng(0)
K, F = 3, 513
for T in (500, 1000):
    act = rng.uniform(size=(K, T)) ** 4
    mask = act[:, None, :] * rng.uniform(0.5, 1.0, size=(K, F, T)) + 0.05 * rng.uniform(size=(K, F, T))
    mask /= mask.sum(0, keepdims=True)
    perm = np.stack([rng.permutation(K) for _ in range(F)], 1)
    m = _lib.to_device(mask[perm, range(F)])
    al = DHTVPermutationAlignment.from_stft_size(1024)
    timed(f'DHTV calculate_mapping K=3 F=513 T={T}', lambda: al.calculate_mapping(m))

from pb_bss_amd import engine
aff = rng.uniform(size=(1, 513, 3, 1000)); aff /= aff.sum(-2, keepdims=True)
affd = _lib.to_device(aff)
timed('estimate_mixture_weight (513, 3, 1000) -> (3, 1000), classes summed over the bins',
      lambda: engine.estimate_mixture_weight(affd, None, True, False))

wv = _lib.to_device(rng.standard_normal((513, 8)) + 1j * rng.standard_normal((513, 8)))
timed('phase_correction (513, 8)', lambda: ex.phase_correction(wv))

Y, init = synth.make_stft(257, 800, 6, 3, seed=0)
y, g0 = _lib.to_device(Y), _lib.to_device(init)
timed('CWMM configs[3] F=257 T=800 D=6 K=3, 100 iterations + predict',
      lambda: CWMMTrainer().fit_predict(y, initialization=g0, iterations=100))
Y, init = synth.make_stft(513, 500, 12, 8, seed=1)
y, g0 = _lib.to_device(Y), _lib.to_device(init)
timed('CWMM generic F=513 T=500 D=12 K=8, 20 iterations + predict',
      lambda: CWMMTrainer().fit_predict(y, initialization=g0, iterations=20))
Yj, e, initj = synth.make_joint(513, 500, 12, 3, 40, seed=2)
yj, ej, gj = _lib.to_device(Yj), _lib.to_device(e), _lib.to_device(initj)
timed('GCACGMM generic F=513 T=500 D=12 K=3 E=40, 20 iterations + predict',
      lambda: GCACGMMTrainer().fit_predict(yj, ej, initialization=gj, iterations=20))
timed('GCACGMM generic, inline permutation alignment, 10 iterations',
      lambda: GCACGMMTrainer().fit_predict(yj, ej, initialization=gj, iterations=10,
                                           weight_constant_axis=(-3,),
                                           inline_permutation_alignment=True), reps=2)
Y, init = synth.make_stft(513, 500, 29, 3, seed=3)
y, g0 = _lib.to_device(Y), _lib.to_device(init)
masks = None


def em29():
    global masks
    masks = CACGMMTrainer().fit_predict(y, initialization=g0, iterations=20)


timed('cACGMM generic F=513 T=500 D=29 K=3, 20 iterations + predict (side chain for bin 512)', em29)
x = y.transpose(-1, -2).contiguous()


def chain29():
    psd = ex.get_power_spectral_density_matrix(x, masks)
    w = ex.get_bf_vector('gev+ban', psd[:, 0], psd[:, 1] + psd[:, 2])
    return ex.apply_beamforming_vector(w, x)


timed('PSD + gev+ban + apply at D=29', chain29)
