#!/usr/bin/env python3
"""EM kernel time for the bin counts rank 0 sees in bench.py at N = 1, 2, 4, 8 GPUs
(N utterances x its block of the 513 bins: 513, 514, 516, 520) and the other ranks' 512."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing.synth import make_stft
Y, init = make_stft(513, 500, 8, 3, seed=0)
Y = np.concatenate([Y, Y]); init = np.concatenate([init, init])
engine.set_timing(True)
for nb in (512, 513, 514, 516, 520, 521, 528):
    y, g = _lib.to_device(Y[:nb]), _lib.to_device(init[:nb])
    ts = []
    for _ in range(5):
        engine.em_fit(y, 3, gamma0=g, iterations=100, final_predict=True, check_status=False)
        ts.append(engine.last_kernel_ms())
    print(f'bins {nb}: kernel {min(ts):.3f} ms', flush=True)
