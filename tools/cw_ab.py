import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import synth
from pb_bss_amd import _lib, engine
from pb_bss_amd.distribution import ComplexWatsonTrainer
F, T, D, K = int(os.environ.get('CW_F', 256)), 800, 6, 3
Y, init = synth.make_stft(F, T, D, K, seed=0)
y, g = _lib.to_device(Y), _lib.to_device(init)
sp = ComplexWatsonTrainer(D).device_spline()
engine.set_timing(True)
best = 1e9
for _ in range(40):
    engine.cwmm_fit(y, K, sp, gamma0=g, iterations=100, final_predict=True, check_status=False)
    best = min(best, engine.last_kernel_ms())
print(os.environ.get('PBBSS_LIB', 'shipped'), 'F', F, 'ms per fit (min of 40) %.4f' % best)
