import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
which = sys.argv[1]
import torch
if which == 'init':
    torch.cuda.init()
elif which == 'set_device':
    torch.cuda.set_device(0)
elif which == 'stream':
    torch.cuda.current_stream(0)
elif which == 'empty':
    torch.empty(1, device='cuda')
elif which == 'count':
    torch.cuda.device_count()
from pb_bss_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
h = ctypes.c_void_p()
print(which, '-> pbbss_create rc', lib.pbbss_create(ctypes.byref(h), 0))
with open('/proc/self/maps') as f:
    libs = sorted({l.split()[-1] for l in f if 'amdhip64' in l or 'hsa-runtime' in l})
print('   ', libs)
