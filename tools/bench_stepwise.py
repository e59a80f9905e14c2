#!/usr/bin/env python3
"""cACGMM fits whose options couple the frequency bins next to the fused fit: F=513 T=500 D=8
K=3, weight_constant_axis (-3,) / (-3, -1) in the cooperative kernel (pbbss_cacgmm_fit_shared)
and in the step-wise loop, and the step-wise loop with the device DHTV aligner inside.
Wall time per EM iteration (torch.cuda.synchronize around the fit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
from pb_bss_amd import _lib
from pb_bss_amd.distribution import CACGMMTrainer
from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment

F, T, D, K, iters = 513, 500, 8, 3, 50
Y, init = synth.make_stft(F, T, D, K, seed=0)
y, g = _lib.to_device(Y), _lib.to_device(init)


def run(label, **kw):
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        CACGMMTrainer().fit(y, initialization=g, iterations=iters, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f'{label}: {best / iters * 1e6:.1f} us per EM iteration ({iters / best:.0f} it/s)')


INLINE_ONLY = '--inline-only' in sys.argv
if INLINE_ONLY:
    from pb_bss_amd import engine
    iters = 20
    al = DHTVPermutationAlignment.from_stft_size(1024)
    run('step-wise, (-3,) + inline device DHTV aligner, iterations 1-20 from a random start',
        weight_constant_axis=(-3,), inline_permutation_aligner=al)
    # the same loop once the EM has settled: resumed from the model after 40 iterations (a
    # handful of bins still flip per iteration on this noisy synthetic mixture)
    model = CACGMMTrainer().fit(y, initialization=g, iterations=40, weight_constant_axis=(-3,),
                                inline_permutation_aligner=al)
    g = model
    run('step-wise, (-3,) + inline device DHTV aligner, iterations 41-60',
        weight_constant_axis=(-3,), inline_permutation_aligner=al)
    sys.exit(0)
run('fused, weight_constant_axis=(-1,)')
run('cooperative, weight_constant_axis=(-3,)', weight_constant_axis=(-3,))
run('cooperative, weight_constant_axis=(-3, -1)', weight_constant_axis=(-3, -1))
from pb_bss_amd import engine
engine.em_fit_shared = lambda *a, **k: None  # force the step-wise loop
run('step-wise, weight_constant_axis=(-3,)', weight_constant_axis=(-3,))
run('step-wise, weight_constant_axis=(-3, -1)', weight_constant_axis=(-3, -1))
run('step-wise, (-3,) + inline device DHTV aligner', weight_constant_axis=(-3,),
    inline_permutation_aligner=DHTVPermutationAlignment.from_stft_size(1024))
