#!/usr/bin/env python3
"""Where does the time between two EM launches go?  Wall clock per step of back-to-back fits with
and without the library's timing events, with (513 bins) and without (512) the side-stream split
kernel, float64 and packed-FP32."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing import synth


def main():
    Y, g = synth.make_stft(513, 500, 8, 3, seed=0)
    for prec in ('f64', 'f32'):
        for nb in (512, 513):
            y, g0 = _lib.to_device(Y[:nb]), _lib.to_device(g[:nb])
            for timing in (True, 'lagged', False):
                engine.set_timing(bool(timing))
                for _ in range(300):
                    engine.em_fit(y, 3, gamma0=g0, iterations=100, final_predict=True,
                                  check_status=False, precision=prec)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 400
                kms = 0.0
                for i in range(n):
                    engine.em_fit(y, 3, gamma0=g0, iterations=100, final_predict=True,
                                  check_status=False, precision=prec)
                    if timing is True:
                        kms += engine.last_kernel_ms()          # waits for THIS launch
                    elif timing == 'lagged' and i >= 2:
                        kms += engine.last_kernel_ms(lag=2) * n / (n - 2)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                print(f'{prec} bins {nb} timing_events {timing}: {dt:.4f} ms per step'
                      + (f', kernel (events) {kms / n:.4f} ms, gap {1e3 * (dt - kms / n):.1f} us'
                         if timing else ''))


if __name__ == '__main__':
    main()
