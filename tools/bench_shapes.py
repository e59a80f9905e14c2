#!/usr/bin/env python3
"""EM throughput across utterance lengths and class counts (F=513, D=8): where the LDS-resident
kernel hands over to the spill variant (observation in an L2 / HBM scratch slab) and what that
costs.  Prints kernel milliseconds per 100-iteration fit and nanoseconds per (bin, frame,
iteration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd.testing import synth
from pb_bss_amd import _lib, engine

engine.set_timing(True)
F, D, iters = 513, 8, 100
for K, T in [(3, 250), (3, 500), (3, 750), (3, 1000), (3, 1500), (3, 2000), (3, 4000), (2, 500), (4, 500), (5, 500), (6, 500)]:
    Y, init = synth.make_stft(F, T, D, K, seed=T + K)
    y, g = _lib.to_device(Y), _lib.to_device(init)
    for _ in range(2):
        engine.em_fit(y, K, gamma0=g, iterations=iters, final_predict=True)
        ms = engine.last_kernel_ms()
    print(f'K={K} T={T:5d}: {ms:8.3f} ms per {iters}-iteration fit, {ms * 1e6 / (F * T * iters):6.3f} ns per bin-frame-iteration, '
          f'{iters / ms * 1e3:8.0f} EM it/s')
