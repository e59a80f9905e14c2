#!/usr/bin/env python3
"""Kernel time of the fused cACGMM fit over the compiled (D, K) range (F=512, T=500,
100 iterations): a quick check that no instantiation is pathologically slow (spills)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import synth
from pb_bss_amd import _lib, engine
engine.set_timing(True)
F, T = 512, 500
print('D K  ms/100it  us/it  (complex64 input)')
for D in (2, 3, 4, 6, 8):
    for K in (1, 2, 3, 4, 5, 6):
        Y, init = synth.make_stft(F, T, D, K, seed=1)
        y, g = _lib.to_device(Y), _lib.to_device(init)
        ms = []
        for _ in range(4):
            engine.em_fit(y, K, gamma0=g, iterations=100, final_predict=True, check_status=False)
            ms.append(engine.last_kernel_ms())
        print(f'{D} {K}  {min(ms):7.3f}  {min(ms) * 10:6.2f}')
