#!/usr/bin/env python3
"""How does the EM kernel time depend on the number of bins around the
256-CU boundary?  (single-utterance tail effect)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
from pb_bss_amd import _lib, engine
Y, init = synth.make_stft(513, 500, 8, 3, seed=0)
Y = np.concatenate([Y, Y]); init = np.concatenate([init, init])
engine.set_timing(True)
BINS = [int(x) for x in sys.argv[1:]] or [256, 384, 500, 512, 513, 514, 600, 700, 768, 769, 1026]
for nb in BINS:
    y, g = _lib.to_device(Y[:nb]), _lib.to_device(init[:nb])
    ts = []
    for _ in range(4):
        engine.em_fit(y, 3, gamma0=g, iterations=100, final_predict=True, check_status=False)
        ts.append(engine.last_kernel_ms())
    print(f'bins {nb}: kernel {min(ts):.3f} ms  ({min(ts)*10:.2f} us/iteration)')
