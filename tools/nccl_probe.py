#!/usr/bin/env python3
"""Development probe: does the per-step mask all-gather (RCCL, same stream) change the
duration of the NEXT persistent EM kernel?  Single GPU, world size 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from oracle import synth
from pb_bss_amd import _lib, engine
from pb_bss_amd.sharding import all_gather_bins
Y, init = synth.make_stft(513, 500, 8, 3, seed=0)
y, g = _lib.to_device(Y), _lib.to_device(init)
engine.set_timing(True)
def t(label, after=None):
    ms = []
    for _ in range(10):
        r = engine.em_fit(y, 3, gamma0=g, iterations=100, final_predict=True, check_status=False)
        ms.append(engine.last_kernel_ms())
        if after:
            after(r['affiliation'])
    torch.cuda.synchronize()
    print(f'{label}: kernel {sum(ms[3:]) / 7:.3f} ms (min {min(ms):.3f})', flush=True)
t('plain')
t('plain + clone of the masks', lambda m: m.clone())
t('plain + 3 clones', lambda m: (m.clone(), m.clone(), m.clone()))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
t('nccl initialised')
t('nccl + all_gather_bins', lambda m: all_gather_bins(m.reshape(1, 513, 3, 500), 513, bin_axis=1))
out = torch.empty(513 * 3 * 500, dtype=torch.float64, device='cuda')
t('nccl + bare all_gather_into_tensor', lambda m: dist.all_gather_into_tensor(out, m.reshape(-1)))
t('nccl, no collective again')
dist.destroy_process_group()
