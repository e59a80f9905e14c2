import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pb_bss_amd import _lib, engine
from pb_bss_amd.distribution import CWMMTrainer
from pb_bss_amd.testing import synth
from oracle import cwmm as ow
F,T,D,K = 257,800,6,3
Y, init = synth.make_stft(F,T,D,K, seed=3)
y, g0 = _lib.to_device(Y), _lib.to_device(init)
for axis in ((-3,-1),):
    tr = CWMMTrainer()
    m = tr.fit(y, initialization=g0, iterations=20, weight_constant_axis=axis)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): m = tr.fit(y, initialization=g0, iterations=20, weight_constant_axis=axis)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print('watson', axis, 'ms per iteration', dt*1e3/20)
    Y128 = Y[:40].astype(np.complex128)
ref = ow.cwmm_fit(Y.astype(np.complex128), init, iterations=6, weight_constant_axis=(-3,-1))
m = CWMMTrainer().fit(Y, initialization=init, iterations=6, weight_constant_axis=(-3,-1))
print('weight err', np.abs(m.weight - ref['weight']).max(), 'conc err', np.abs(m.complex_watson.concentration-ref['concentration']).max(), m.weight.shape, ref['weight'].shape)
