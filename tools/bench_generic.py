#!/usr/bin/env python3
"""Timing of the generic-size path (9 <= D <= 32 sensors): fused-call fit_predict of the cACGMM
(F=513, T=500, K=3, 20 iterations) + PSD + gev+ban, next to the NumPy oracle (bounded sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cacgmm as oc, synth
from pb_bss_amd import _lib, engine
from pb_bss_amd import extraction as ex

F, T, K, iters = 513, 500, 3, 20
engine.set_timing(True)
for D in (12, 16, 24, 29):  # apply_beamforming_vector asserts D < 30 like the reference
    Y, init = synth.make_stft(F, T, D, K, seed=D)
    y, g = _lib.to_device(Y), _lib.to_device(init)
    for _ in range(2):
        r = engine.em_fit(y, K, gamma0=g, iterations=iters, final_predict=True, check_status=False)
        ms = engine.last_kernel_ms()
    X = y.transpose(1, 2).contiguous()
    for rep in range(2):  # the first call of a process loads torch's / the library's code objects
        torch.cuda.synchronize(); t0 = time.perf_counter()  # (tools/first_call_probe.py)
        psd = ex.get_power_spectral_density_matrix(X, r['affiliation'])
        w = ex.get_bf_vector('gev+ban', psd[:, 0], psd[:, 1] + psd[:, 2])
        s = ex.apply_beamforming_vector(w, X)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # useful float64 flops per frame and EM iteration of the generic-size path: Hermitian quadratic
    # form y^H A_k y (4 D^2 per class), outer product of the M-step (3 D^2), C_k += w_k P (2 D^2 per
    # class) -> D^2 (6 K + 3); bytes: one read of the complex64 observation per E-step and per M-step
    flops = D * D * (6 * K + 3) * F * T * iters
    tf = flops / (ms * 1e-3) / 1e12
    gbs = 2 * 8.0 * F * T * D * iters / (ms * 1e-3) / 1e9
    line = (f'D={D}: {iters} EM iterations in {ms:.2f} ms -> {iters / ms * 1e3:.0f} EM it/s '
            f'({ms / iters * 1e3:.0f} us/iter; {tf:.1f} TFLOP/s = {tf / 78.6 * 100:.1f} % of the FP64 vector '
            f'peak, {gbs:.0f} GB/s of observation reads = {gbs / 8000 * 100:.1f} % of 8 TB/s); '
            f'psd + gev+ban + apply {dt * 1e3:.2f} ms')
    if '--no-cpu' not in sys.argv and D in (16,):
        Y128 = Y.astype(np.complex128)
        t0 = time.perf_counter(); oc.em_fit(Y128[:64], init[:64], iterations=2); dt = time.perf_counter() - t0
        line += f'; NumPy oracle {2 / dt * 64 / F:.2f} EM it/s (64-bin sample scaled to {F} bins)'
    print(line)
