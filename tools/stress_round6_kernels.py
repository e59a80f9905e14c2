#!/usr/bin/env python3
"""Random-shape parity sweep of the round-6 kernels against the oracle (development aid, not part of
the suite: `gpurun -- 'python tools/stress_round6_kernels.py [trials]'`).  vMF bin kernel
(vmf_bin.hip), Watson kernels (eight-wave / four-wave / split, cwmm.hpp), the wave-private joint
sweep (embed.hip).  Prints one line per case and the number of cases outside the tolerance."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cwmm as ocw, embed as oe, synth  # noqa: E402
from pb_bss_amd.distribution import (CWMMTrainer, GCACGMMTrainer, VMFCACGMMTrainer,  # noqa: E402
                                     VMFMMTrainer)

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(2026)
bad = 0


def clustered(B, N, E, K):
    mu = rng.standard_normal((B, K, E))
    lab = rng.integers(K, size=(B, N))
    y = np.take_along_axis(mu, lab[..., None], 1) + 0.6 * rng.standard_normal((B, N, E))
    init = rng.uniform(size=(B, K, N)) + 2.0 * (np.arange(K)[None, :, None] == lab[:, None, :])
    return y, init / init.sum(1, keepdims=True)


for _ in range(trials):
    B = int(rng.choice([16, 31, 200, 256, 257, 300, 520]))
    N, E, K = int(rng.integers(20, 900)), int(rng.integers(2, 17)), int(rng.integers(2, 5))
    y, init = clustered(B, N, E, K)
    y = y.astype(np.float32)
    sal = rng.uniform(0.1, 1.0, size=(B, N)) if rng.uniform() < 0.5 else None
    ref = oe.vmfmm_fit(y.astype(np.float64), init, 6, saliency=sal)
    got = VMFMMTrainer().fit_predict(y, initialization=init, iterations=6, saliency=sal)
    err = np.abs(got - oe.vmfmm_predict(ref, y.astype(np.float64))).max()
    print(f'vmf    B={B:4d} N={N:4d} E={E:2d} K={K} sal={sal is not None!s:5} err={err:.2e}', flush=True)
    bad += not err < 1e-7

for _ in range(trials):
    F = int(rng.choice([5, 100, 256, 257, 258, 300, 513]))
    T, D, K = int(rng.integers(70, 900)), int(rng.integers(2, 9)), int(rng.integers(2, 5))
    lab = rng.integers(K, size=(F, T))
    a = rng.standard_normal((F, K, D)) + 1j * rng.standard_normal((F, K, D))
    s = rng.standard_normal((F, T)) + 1j * rng.standard_normal((F, T))
    Y = np.take_along_axis(a, lab[..., None], 1) * s[..., None]
    Y = (Y + 0.3 * (rng.standard_normal(Y.shape) + 1j * rng.standard_normal(Y.shape))).astype(np.complex64)
    init = rng.uniform(size=(F, K, T)) + 2.0 * (np.arange(K)[None, :, None] == lab[:, None, :])
    init /= init.sum(1, keepdims=True)
    ref = ocw.cwmm_fit(Y.astype(np.complex128), init, 6)
    got = CWMMTrainer().fit_predict(Y, initialization=init, iterations=6)
    err = np.abs(got - ocw.cwmm_predict(ref, Y.astype(np.complex128))).max()
    print(f'watson F={F:4d} T={T:4d} D={D} K={K} err={err:.2e}', flush=True)
    bad += not err < 1e-6

for _ in range(trials):
    F, T = int(rng.choice([1, 3, 40, 130])), int(rng.integers(64, 400))
    D, K, E = int(rng.integers(2, 9)), int(rng.integers(2, 5)), int(rng.choice([4, 8, 12, 20, 40, 44, 60]))
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=int(rng.integers(1 << 30)))
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    sal = rng.uniform(0.1, 1.0, size=(F, T)) if rng.uniform() < 0.5 else None
    for kind, trainer, kw in (('gaussian', GCACGMMTrainer(), {}),
                              ('vmf', VMFCACGMMTrainer(), dict(max_concentration=80.))):
        got = trainer.fit_predict(Y, e, initialization=init, iterations=4, saliency=sal, **kw)
        ref = oe.joint_fit(kind, Y128, e64, init, 4, saliency=sal, **kw)
        err = np.abs(got - oe.joint_model_predict(ref, Y128, e64)).max()
        print(f'joint  {kind:8s} F={F:3d} T={T:3d} D={D} K={K} E={E:2d} sal={sal is not None!s:5} '
              f'err={err:.2e}', flush=True)
        bad += not err < 1e-6
print('cases outside the tolerance:', bad)
