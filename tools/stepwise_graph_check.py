#!/usr/bin/env python3
"""Step-wise EM loop launch by launch vs as ONE captured graph per iteration (round 5):
same results bit for bit, wall time per EM iteration.  F=513 T=500 D=8 K=3, 30 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
from pb_bss_amd import _lib, engine
from pb_bss_amd.distribution import CACGMMTrainer
from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment

F, T, D, K, iters = 513, 500, 8, 3, 30
Y, init = synth.make_stft(F, T, D, K, seed=0)
y, g = _lib.to_device(Y), _lib.to_device(init)


def fit(graph, **kw):
    os.environ['PBBSS_STEPWISE_GRAPH'] = '1' if graph else '0'
    best, m = None, None
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = CACGMMTrainer().fit(y, initialization=g, iterations=iters, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return m, best / iters * 1e6


def compare(label, **kw):
    a, ta = fit(False, **kw)
    b, tb = fit(True, **kw)
    same = (np.array_equal(_lib.to_host(a.weight), _lib.to_host(b.weight)) and
            np.array_equal(_lib.to_host(a.cacg.covariance_eigenvalues),
                           _lib.to_host(b.cacg.covariance_eigenvalues)) and
            np.array_equal(_lib.to_host(a.cacg.covariance_eigenvectors),
                           _lib.to_host(b.cacg.covariance_eigenvectors)))
    print(f'{label}: eager {ta:.1f} us / iteration, graph {tb:.1f} us / iteration, identical: {same}')


al = DHTVPermutationAlignment.from_stft_size(1024)
compare('(-3,) + inline device DHTV aligner', weight_constant_axis=(-3,), inline_permutation_aligner=al)
engine.em_fit_shared = lambda *a, **k: None  # force the step-wise loop for the plain coupled weights
compare('(-3,) step-wise', weight_constant_axis=(-3,))
compare('(-3, -1) step-wise', weight_constant_axis=(-3, -1))
