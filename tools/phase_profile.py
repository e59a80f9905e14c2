#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of the EM kernel (needs the profiling build:
`make -C pb_bss_amd/csrc prof`, run with PBBSS_LIB=libpbbss_hip_prof.so)."""
import os
import sys
import ctypes
os.environ.setdefault('PBBSS_LIB', 'libpbbss_hip_prof.so')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import synth
from pb_bss_amd import _lib, engine

NAMES = ['load/init', 'E', 'E-barrier+sums', 'M', 'M-barrier', 'factor', 'factor-barrier', 'final predict']


def main():
    f32 = '--f32' in sys.argv
    if f32:
        sys.argv.remove('--f32')
    kw = {'precision': 'f32'} if f32 else {}
    nutt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 513 * nutt  # bins actually used (e.g. 512: no tail)
    iters = 100
    Y = np.concatenate([synth.make_stft(513, 500, 8, 3, seed=s)[0] for s in range(nutt)])
    g = np.concatenate([synth.make_stft(513, 500, 8, 3, seed=s)[1] for s in range(nutt)])
    Y, g = Y[:nb], g[:nb]
    y, g0 = _lib.to_device(Y), _lib.to_device(g)
    cnt = torch.zeros(72, dtype=torch.int64, device='cuda')
    h = _lib.handle()
    _lib.load().pbbss_set_phase_profile(h, ctypes.c_void_p(cnt.data_ptr()))
    engine.set_timing(True)
    engine.em_fit(y, 3, gamma0=g0, iterations=iters, final_predict=True, **kw)  # warm-up
    cnt.zero_()
    engine.em_fit(y, 3, gamma0=g0, iterations=iters, final_predict=True, **kw)
    ms = engine.last_kernel_ms()
    call = cnt.cpu().numpy().astype(np.float64)
    c = call[:32].reshape(4, 8)
    cs = call[32:64].reshape(4, 8)
    if call[70] > 0:
        print(f'  phase F (class 0, fast path): sums+scale+load {call[67]/call[70]:.0f}, trace+Gauss-Jordan {call[68]/call[70]:.0f}, bound+store {call[69]/call[70]:.0f} ticks per call')
    if call[66] > 0:
        print(f'  phase M (wave 0): butterfly {call[64]/call[66]:.0f} ticks, write-back {call[65]/call[66]:.0f} ticks per call')
    nwg = min(nb, 256 * 3)
    print(f'utterances {nutt}: kernel {ms:.3f} ms, {nutt*iters/ms*1e3:.0f} utt-iter/s; cycles per workgroup-iteration '
          f'(avg over {nwg} workgroups, s_memtime ticks):')
    per = c / nb / iters  # per problem-iteration
    for w in range(4):
        print(f'  wave {w}: ' + ', '.join(f'{n} {per[w, i]:.0f}' for i, n in enumerate(NAMES)))
    tot = per[0].sum()
    if cs.sum() > 0:
        G = (500 + 63) // 64
        print('  split group (per workgroup-iteration, 8 workgroups): 1 E, 2 E-barrier, 3 M, 4 M-barrier+exchange, 5 factor, 6 barrier')
        for w in range(4):
            print(f'    wave {w}: ' + ', '.join(f'{i}:{cs[w, i] / G / iters:.0f}' for i in range(8)))
    print(f'  wave0 total {tot:.0f} ticks per problem-iteration')


if __name__ == '__main__':
    main()
