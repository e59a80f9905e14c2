#!/usr/bin/env python3
"""Persistent EM kernel, many rounds per workgroup: does the time per round of 768 bins stay at the
single-round figure, or does the sustained clock drop further on long launches?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing.synth import make_stft
Y, init = make_stft(513, 500, 8, 3, seed=0)
engine.set_timing(True)
for nb in (512, 768, 1536, 3072, 7680, 15360, 32832):
    reps = -(-nb // 513)
    y = _lib.to_device(np.concatenate([Y] * reps)[:nb])
    g = _lib.to_device(np.concatenate([init] * reps)[:nb])
    ts = []
    for _ in range(3):
        engine.em_fit(y, 3, gamma0=g, iterations=100, final_predict=True, check_status=False)
        ts.append(engine.last_kernel_ms())
    rounds = nb / 768.0
    print(f'bins {nb}: kernel {min(ts):.3f} ms  rounds {rounds:.2f}  {min(ts)/max(rounds,1):.3f} ms/round  '
          f'{nb/513*100/min(ts):.1f}k utt-it/s', flush=True)
