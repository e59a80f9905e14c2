"""Scratch diagnostics for a gpurun session (not a test)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import synth, cacgmm as oc
from pb_bss_amd import engine, _lib
import torch

def run(Y, init, it, **kw):
    r = engine.em_fit(_lib.to_device(Y), init.shape[1], gamma0=_lib.to_device(init), iterations=it, final_predict=True, **kw)
    return _lib.to_host(r['affiliation'])

for name, dtype, zero in [('c64', np.complex64, False), ('c64+zero', np.complex64, True), ('c128', np.complex128, False), ('c128+zero', np.complex128, True)]:
    Y, init = synth.make_stft(7, 150, 4, 2, seed=5, dtype=dtype)
    if zero: Y[:, 10] = 0
    Y128 = Y.astype(np.complex128)
    for it in (1, 4):
        m = oc.em_fit(Y128, init, iterations=it); mask = oc.em_predict(m, Y128)
        d = run(Y, init, it)
        e = np.abs(d - mask)
        print(name, 'it', it, 'err', e.max(), 'argmax', np.unravel_index(e.argmax(), e.shape), 'err excluding t=10', np.delete(e, 10, axis=-1).max())

# timing: config 2
Y, init = synth.make_stft(513, 500, 8, 3, seed=0)
y = _lib.to_device(Y); g = _lib.to_device(init)
engine.set_timing(True)
for iters in (1, 10, 100, 100, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = engine.em_fit(y, 3, gamma0=g, iterations=iters, final_predict=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = engine.last_kernel_ms()
    print(f'config2 iters={iters}: wall {1e3*(t1-t0):.3f} ms, kernel {ms:.3f} ms, {iters/ms*1e3:.0f} iter/s, slowpath={(int((r["status"]&8).sum()))}')
# batch of 16 utterances
Yb = np.concatenate([synth.make_stft(513, 500, 8, 3, seed=s)[0] for s in range(16)]); ib = np.concatenate([synth.make_stft(513, 500, 8, 3, seed=s)[1] for s in range(16)])
y = _lib.to_device(Yb); g = _lib.to_device(ib)
for iters in (100, 100):
    r = engine.em_fit(y, 3, gamma0=g, iterations=iters, final_predict=False)
    ms = engine.last_kernel_ms()
    print(f'batch16 iters={iters}: kernel {ms:.3f} ms, {16*iters/ms*1e3:.0f} utt-iter/s')
