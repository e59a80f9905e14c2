import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, warnings
from pb_bss_amd import _lib, engine
from pb_bss_amd.testing import synth
for (F,T,D,K) in ((257,700,8,3),(513,500,8,3)):
    Y, init = synth.make_stft(F,T,D,K, seed=33)
    y, g0 = _lib.to_device(Y), _lib.to_device(init)
    for lim in (0, 1):
        engine.split_reset(); engine.set_spin_limit(lim)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            r = engine.em_fit_shared(y, K, F, weight_mode=_lib.WEIGHT_SHARED_K, gamma0=g0, iterations=5, final_predict=True, check_status=False)
        torch.cuda.synchronize()
        print(F,T,'limit',lim,'r is None',r is None, 'status or', None if r is None else int(np.bitwise_or.reduce(_lib.to_host(r['status']).ravel())), 'split_error', engine.split_error(), [str(x.message)[:60] for x in w])
engine.set_spin_limit(0); engine.split_reset()
