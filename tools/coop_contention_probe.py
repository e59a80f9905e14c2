#!/usr/bin/env python3
"""How often is the cooperative shared-weight fit "not served" (its grid barrier timed out) while
other EM fits of the SAME process run beside it?  The scenario of tests/test_gpu_contention.py,
counted over many rounds: three host threads (own handle and stream each) start together, one
loops over 6 cooperative fits (F=513 T=500 D=8 K=3, 12 iterations), the others over 6 float64
(split groups) and 6 packed-FP32 (in-grid members) fits.  A/B of the residency gate:
    PBBSS_RESIDENCY_GATE=1 python tools/coop_contention_probe.py [rounds]     (default)
    PBBSS_RESIDENCY_GATE=0 python tools/coop_contention_probe.py [rounds]"""
import os, sys, threading, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
F, T, D, K, iters, reps = 513, 500, 8, 3, 12, 6
Y, init = synth.make_stft(F, T, D, K, seed=0)
Y2, init2 = synth.make_stft(F, T, D, K, seed=1)
y, g0 = _lib.to_device(Y), _lib.to_device(init)
y2, g2 = _lib.to_device(Y2), _lib.to_device(init2)
count = {'served': 0, 'not_served': 0}


def coop():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        for _ in range(reps):
            r = engine.em_fit_shared(y2, K, F, weight_mode=_lib.WEIGHT_SHARED_K, gamma0=g2,
                                     iterations=iters, final_predict=True)
            count['served' if r is not None else 'not_served'] += 1
        s.synchronize()


def other(precision):
    s = torch.cuda.Stream()
    kw = {} if precision == 'f64' else {'precision': 'f32'}
    with torch.cuda.stream(s):
        for _ in range(reps):
            engine.em_fit(y, K, gamma0=g0, iterations=iters, final_predict=True, **kw)
        s.synchronize()


t0 = time.perf_counter()
for _ in range(R):
    th = [threading.Thread(target=other, args=('f64',)), threading.Thread(target=coop),
          threading.Thread(target=other, args=('f32',))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
print(f"PBBSS_RESIDENCY_GATE={os.environ.get('PBBSS_RESIDENCY_GATE', '1')}: cooperative fits served "
      f"{count['served']} / {R * reps}, not served {count['not_served']}; {time.perf_counter() - t0:.1f} s")
