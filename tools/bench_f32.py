#!/usr/bin/env python3
"""Kernel time of the packed-FP32 EM kernel on the headline shape (A/B of library builds on one
box: PBBSS_LIB=<variant>.so python tools/bench_f32.py [bins] [reps])."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing import synth


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 513
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    prec = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    Y, g = synth.make_stft(513, 500, 8, 3, seed=0)
    y, g0 = _lib.to_device(Y[:nb]), _lib.to_device(g[:nb])
    engine.set_timing(True)
    if os.environ.get('NOSPLIT'):
        engine.set_split_tail(False)
    for _ in range(300):  # clock ramp
        engine.em_fit(y, 3, gamma0=g0, iterations=100, final_predict=True, precision=prec,
                      check_status=False)
    ms = []
    for _ in range(reps):
        engine.em_fit(y, 3, gamma0=g0, iterations=100, final_predict=True, precision=prec,
                      check_status=False)
        ms.append(engine.last_kernel_ms())
    ms = np.array(ms)
    print(f'{os.environ.get("PBBSS_LIB", "libpbbss_hip.so"):28s} {prec} bins {nb}: kernel median '
          f'{np.median(ms):.4f} ms  min {ms.min():.4f}  ({nb / 513 * 1e5 / np.median(ms):.0f} it/s)')


if __name__ == '__main__':
    main()
