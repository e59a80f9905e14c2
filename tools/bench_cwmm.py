#!/usr/bin/env python3
"""Timing of the complex-Watson mixture EM at BASELINE config 4 (F=257, T=800,
D=6, K=3, 100 iterations) next to the NumPy oracle (bounded sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cwmm as ow, synth
from pb_bss_amd import _lib, engine
from pb_bss_amd.distribution import ComplexWatsonTrainer

F, T, D, K = 257, 800, 6, 3
Y, init = synth.make_stft(F, T, D, K, seed=0)
y, g = _lib.to_device(Y), _lib.to_device(init)
sp = ComplexWatsonTrainer(D).device_spline()
engine.set_timing(True)
for iters in (100, 100, 100):
    engine.cwmm_fit(y, K, sp, gamma0=g, iterations=iters, final_predict=True)
    ms = engine.last_kernel_ms()
    print(f'device cWMM config 4: {iters} iterations in {ms:.3f} ms -> {iters/ms*1e3:.0f} EM it/s')
t0 = time.perf_counter(); ow.cwmm_fit(Y.astype(np.complex128), init, iterations=10); dt = time.perf_counter() - t0
print(f'NumPy oracle: {10/dt:.1f} EM it/s')
