#!/usr/bin/env python3
"""Timing of the device DHTV permutation alignment vs the NumPy oracle
(K=3, F=513, T=500: the masks of BASELINE config 2; PBBSS_BENCH_T=1000: config 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import permutation_alignment as op
from pb_bss_amd import _lib, engine
from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment

rng = np.random.default_rng(0)
K, F = 3, 513
T = int(os.environ.get('PBBSS_BENCH_T', '500'))
act = rng.uniform(size=(K, T)) ** 4
mask = act[:, None, :] * rng.uniform(0.5, 1.0, size=(K, F, T)) + 0.05 * rng.uniform(size=(K, F, T))
mask /= mask.sum(0, keepdims=True)
perm = np.stack([rng.permutation(K) for _ in range(F)], 1)
pm = mask[perm, range(F)]
solver = DHTVPermutationAlignment.from_stft_size(1024)
plan = _lib.to_device(np.asarray(solver.alignment_plan, np.int32))
for U in (1, 2, 4, 8, 16, 64):
    m = _lib.to_device(np.broadcast_to(pm, (U, K, F, T)).copy())
    engine.dhtv_calculate_mapping(m, plan)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        mapping, feat, st = engine.dhtv_calculate_mapping(m, plan)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'device DHTV U={U}: {dt*1e3:.3f} ms per call, {dt/U*1e3:.3f} ms/utterance')
t0 = time.perf_counter(); ref = op.dhtv_calculate_mapping(pm, solver.alignment_plan); dt = time.perf_counter() - t0
print(f'NumPy oracle: {dt*1e3:.1f} ms/utterance; identical mapping: {(ref == _lib.to_host(mapping[0])).all()}')

# ---- pairwise solvers (greedy / oracle), every metric ----------------------------------------
from pb_bss_amd.permutation_alignment import GreedyPermutationAlignment, OraclePermutationAlignment
md, rd = _lib.to_device(pm), _lib.to_device(mask)
for metric in ('cos', 'multiply', 'euclidean'):
    g, o = GreedyPermutationAlignment(metric), OraclePermutationAlignment(metric)
    g.calculate_mapping(md); o.calculate_mapping(md, rd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        mg = g.calculate_mapping(md)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        mo = o.calculate_mapping(md, rd)
    torch.cuda.synchronize(); to = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter(); rg = op.greedy_calculate_mapping(pm, metric); tcg = time.perf_counter() - t0
    t0 = time.perf_counter(); ro = op.oracle_calculate_mapping(pm, mask, metric); tco = time.perf_counter() - t0
    print(f'{metric:10s} greedy: device {tg*1e3:.3f} ms, NumPy oracle {tcg*1e3:.1f} ms, identical '
          f'{(rg == mg.cpu().numpy()).all()};  oracle solver: device {to*1e3:.3f} ms, NumPy {tco*1e3:.1f} ms, '
          f'identical {(ro == mo.cpu().numpy()).all()}')
