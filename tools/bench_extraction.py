#!/usr/bin/env python3
"""Timings of the extraction kernels at BASELINE config-2 shapes (F=513, T=500,
D=8, K=3) with data resident on the device, next to the NumPy oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import beamformer as ob, synth
from pb_bss_amd import _lib, engine

F, T, D, K = 513, 500, 8, 3
Y, init = synth.make_stft(F, T, D, K, seed=0)
X = np.ascontiguousarray(Y.transpose(0, 2, 1))
mask = init
x, m = _lib.to_device(X), _lib.to_device(mask)


def gpu_time(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out


def cpu_time(fn):
    t0 = time.perf_counter(); out = fn(); return (time.perf_counter() - t0) * 1e3, out

rows = []
t, psd = gpu_time(lambda: engine.psd(x, m)); c, psd_ref = cpu_time(lambda: ob.psd(X.astype(np.complex128), mask))
rows.append(('psd (F,K,D,D)', t, c, 8.0 * F * D * T / (t * 1e-3) / 1e9))
tp = psd[:, 0].contiguous(); nn = (psd[:, 1] + psd[:, 2]).contiguous()
tpr, nnr = psd_ref[:, 0], psd_ref[:, 1] + psd_ref[:, 2]
t, (w, st) = gpu_time(lambda: engine.gev(tp, nn)); c, wr = cpu_time(lambda: ob.gev_vector(tpr, nnr)); rows.append(('gev', t, c, None))
t, _ = gpu_time(lambda: engine.mvdr_souden(tp, nn, 2.2e-308)); c, _ = cpu_time(lambda: ob.mvdr_souden(tpr, nnr)); rows.append(('mvdr_souden', t, c, None))
t, wb = gpu_time(lambda: engine.ban(w, nn)); c, _ = cpu_time(lambda: ob.ban(wr, nnr)); rows.append(('ban', t, c, None))
t, _ = gpu_time(lambda: engine.apply_bf(wb, x)); c, _ = cpu_time(lambda: ob.apply_bf(wr, X.astype(np.complex128))); rows.append(('apply', t, c, (8.0 * F * D * T + 16.0 * F * T) / (t * 1e-3) / 1e9))
t, _ = gpu_time(lambda: engine.heev(tp)); c, _ = cpu_time(lambda: np.linalg.eigh(tpr)); rows.append(('heev (pca)', t, c, None))
t, _ = gpu_time(lambda: engine.normalize_observation(_lib.to_device(Y))); rows.append(('normalize (incl. H2D)', t, float('nan'), None))
print('kernel | device ms | numpy ms | GB/s (algorithmic)')
for r in rows:
    print(f'{r[0]} | {r[1]:.3f} | {r[2]:.2f} | ' + (f'{r[3]:.0f}' if r[3] else '-'))
