#!/usr/bin/env python3
"""Timing of the full-covariance Gaussian kernels (csrc/gauss_full.hip) at the embedding size of
BASELINE config 5 (N = 513*500 points, E = 40, K = 3): weighted scatter (FP64 MFMA Gram tiles),
log-pdf (MFMA quadratic forms), and the GMM EM loop; NumPy oracle on a bounded sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd import _lib, engine

N, E, K = 513 * 500, 40, 3
rng = np.random.default_rng(0)
centers = rng.normal(size=(K, E)) * 1.5
lab = rng.integers(K, size=N)
y = (centers[lab] + rng.normal(size=(N, E))).astype(np.float32)
init = rng.uniform(size=(K, N)); init /= init.sum(0, keepdims=True)
yd = _lib.to_device(y)[None]
gd = _lib.to_device(init)[None].contiguous()
engine.set_timing(True)
P = 16 * ((E + 1 + 15) // 16)
for rep in range(3):
    mean, cov = engine.gauss_full_fit(yd, gd)
    ms = engine.last_kernel_ms()
flops = 2.0 * (P // 16) * (P // 16 + 1) / 2 * 256 * 4 * (N / 4) * K
print(f'scatter + finalize (fit): {ms*1e3:.1f} us; MFMA tiles {flops/1e9:.2f} GFLOP -> '
      f'{flops/ms/1e9:.1f} TFLOP/s of 78.6 (matrix FP64 peak)')
for rep in range(3):
    lp, st = engine.gauss_full_log_pdf(yd, mean, cov)
    ms = engine.last_kernel_ms()
flops = 2.0 * (P // 16) * (P // 4) * 256 * 4 * (N / 16) * K
print(f'factor + log-pdf: {ms*1e3:.1f} us; MFMA tiles {flops/1e9:.2f} GFLOP -> {flops/ms/1e9:.1f} TFLOP/s of 78.6')
iters = 20
for rep in range(3):
    r = engine.gmm_full_fit(yd, K, gamma0=gd, iterations=iters, final_predict=True)
    ms = engine.last_kernel_ms()
print(f'device GMM (full covariance) N={N} E={E} K={K}: {iters} iterations in {ms:.3f} ms -> '
      f'{iters/ms*1e3:.0f} EM it/s, {ms/iters*1e3:.1f} us/iter, status {int(r["status"].item())}')
for rep in range(3):
    r = engine.gmm_fit(yd, K, gamma0=gd, iterations=iters, final_predict=True)
    ms = engine.last_kernel_ms()
print(f'device GMM (spherical): {ms/iters*1e3:.1f} us/iter')
if '--no-cpu' not in sys.argv:
    from oracle import embed as oe
    y64 = y.astype(np.float64)
    t0 = time.perf_counter(); oe.gmm_fit(y64, init, 2, covariance_type='full'); dt = time.perf_counter() - t0
    print(f'NumPy oracle GMM (full): {2/dt:.2f} EM it/s')
