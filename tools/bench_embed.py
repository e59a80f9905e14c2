#!/usr/bin/env python3
"""Timing of the real-embedding mixtures next to the NumPy oracle (bounded sample):
  * vMF mixture on N = 513*500 Deep-Clustering-style embeddings (E=40, K=3), 100 iterations
  * BASELINE config 5: GCACGMM / VMFCACGMM, F=513 T=500 D=8 K=3 E=40, 100 iterations
Whole-loop device time from HIP events around the enqueued kernels (pbbss_set_timing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import embed as oe, synth
from pb_bss_amd import _lib, engine

F, T, D, K, E = int(os.environ.get('BENCH_F', '513')), 500, 8, 3, 40  # BENCH_F=512: no remainder bin
Y, e, init = synth.make_joint(F, T, D, K, E, seed=0)
yd, ed, gd = _lib.to_device(Y), _lib.to_device(e), _lib.to_device(init)
flat = ed.reshape(1, F * T, E)
g0 = gd.permute(1, 0, 2).reshape(1, K, F * T).contiguous()
engine.set_timing(True)
iters = 100
for rep in range(0 if '--joint-only' in sys.argv else 3):
    engine.vmfmm_fit(flat, K, gamma0=g0, iterations=iters, final_predict=True)
    ms = engine.last_kernel_ms()
    print(f'device vMFMM N={F*T} E={E} K={K}: {iters} iterations in {ms:.3f} ms -> '
          f'{iters/ms*1e3:.0f} EM it/s, {ms/iters*1e3:.1f} us/iter')
for kind, name in ((_lib.EMBED_GAUSS_SPHERICAL, 'GCACGMM'), (_lib.EMBED_VMF, 'VMFCACGMM')):
    for rep in range(3):
        engine.joint_fit(yd, ed, K, kind, gamma0=gd, iterations=iters, final_predict=True)
        ms = engine.last_kernel_ms()
        print(f'device {name} config 5: {iters} iterations in {ms:.3f} ms -> '
              f'{iters/ms*1e3:.0f} EM it/s, {ms/iters*1e3:.1f} us/iter')
if '--no-cpu' not in sys.argv and '--joint-only' not in sys.argv:
    e64, Y128 = e.astype(np.float64), Y.astype(np.complex128)
    t0 = time.perf_counter(); oe.vmfmm_fit(e64.reshape(-1, E), init.transpose(1, 0, 2).reshape(K, -1), 5)
    dt = time.perf_counter() - t0
    print(f'NumPy oracle vMFMM: {5/dt:.2f} EM it/s')
    t0 = time.perf_counter(); oe.joint_fit('gaussian', Y128, e64, init, 3); dt = time.perf_counter() - t0
    print(f'NumPy oracle GCACGMM: {3/dt:.2f} EM it/s')
