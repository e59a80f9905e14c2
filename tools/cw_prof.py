"""Development: per-phase cycle counters of cwmm_em_kernel<6,3> (a library built with
tools/dev_variant_cw.sh prof 6 -DPBBSS_CW_PROF).  One workgroup (block 7) adds the s_memtime
differences of its four waves per EM iteration to a device array."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import synth
from pb_bss_amd import _lib, engine
from pb_bss_amd.distribution import ComplexWatsonTrainer
F, T, D, K = int(os.environ.get('CW_F', 256)), 800, 6, 3
Y, init = synth.make_stft(F, T, D, K, seed=0)
y, g = _lib.to_device(Y), _lib.to_device(init)
sp = ComplexWatsonTrainer(D).device_spline()
engine.set_timing(True)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 64)()
for _ in range(5):
    engine.cwmm_fit(y, K, sp, gamma0=g, iterations=100, final_predict=True, check_status=False)
lib.pbbss_dev_cw_prof(buf, 1)
engine.cwmm_fit(y, K, sp, gamma0=g, iterations=100, final_predict=True, check_status=False)
ms = engine.last_kernel_ms()
import torch; torch.cuda.synchronize()
lib.pbbss_dev_cw_prof(buf, 0)
a = np.array(list(buf), dtype=np.float64)
n = max(a[7], 1)
print(f'kernel {ms:.4f} ms per fit; factor_class of class 0, ticks per call: sums+scale+finite {a[0]/n:.0f}, eigenpair {a[1]/n:.0f}, concentration {a[2]/n:.0f}, set_class (log-norm) {a[3]/n:.0f}')
