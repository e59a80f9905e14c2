#!/usr/bin/env python3
"""Where does the one-time ~200 ms of the first generic-size beamformer call go?
(profiles/r01_f_generic_kernels.txt showed 'D=12 ... psd + gev+ban + apply 203 ms' once, <1 ms
afterwards.)  Times every stage of the first and second call separately, with the torch ops the
wrappers use for their status checks warmed up first or not (--warm-torch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_bss_amd.testing import synth
from pb_bss_amd import _lib, engine
from pb_bss_amd import extraction as ex


def timed(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'  {label:46s} {dt * 1e3:9.2f} ms')
    return out


F, T, K, D = 513, 500, 3, 12
Y, init = synth.make_stft(F, T, D, K, seed=D)
y, g = _lib.to_device(Y), _lib.to_device(init)
r = engine.em_fit(y, K, gamma0=g, iterations=2, final_predict=True, check_status=False)
X = y.transpose(1, 2).contiguous()
if '--warm-torch' in sys.argv:
    print('torch ops used by the status checks, first use:')
    z = torch.zeros(1000, dtype=torch.int32, device='cuda')
    timed('(int32 != 0).any().item()', lambda: bool((z != 0).any().item()))
    timed('int32.max().item()', lambda: int(z.max().item()))
    c = torch.zeros(10, 4, 4, dtype=torch.complex128, device='cuda')
    timed('complex transpose/conj/contiguous', lambda: c.transpose(1, 2).conj().contiguous())
    timed('complex add', lambda: c + c)
for rep in (1, 2):
    print(f'call {rep}:')
    psd = timed('get_power_spectral_density_matrix', lambda: ex.get_power_spectral_density_matrix(X, r['affiliation']))
    noise = timed('noise = psd[:, 1] + psd[:, 2]', lambda: psd[:, 1] + psd[:, 2])
    w = timed("get_bf_vector('gev+ban')", lambda: ex.get_bf_vector('gev+ban', psd[:, 0], noise))
    s = timed('apply_beamforming_vector', lambda: ex.apply_beamforming_vector(w, X))
