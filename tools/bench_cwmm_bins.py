import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import synth
from pb_bss_amd import _lib, engine
from pb_bss_amd.distribution import ComplexWatsonTrainer
T, D, K = 800, 6, 3
sp = ComplexWatsonTrainer(D).device_spline()
engine.set_timing(True)
for F in (128, 256, 257, 512, 513, 768):
    Y, init = synth.make_stft(F, T, D, K, seed=0)
    y, g = _lib.to_device(Y), _lib.to_device(init)
    best = 1e9
    for _ in range(3):
        engine.cwmm_fit(y, K, sp, gamma0=g, iterations=100, final_predict=True)
        best = min(best, engine.last_kernel_ms())
    print(f'F={F}: {best:.3f} ms per 100 iterations')
