#!/usr/bin/env python3
"""1-GPU scaling MODEL of the sharded paths (round 5; no multi-GPU run has ever been available).

    gpurun -- 'python tools/scaling_model.py > gpurun_out/r05_scaling_model.json'

For N = 1, 2, 4, 8 ranks this times, on ONE MI355X, exactly the share of the work that the
slowest rank of an N-rank run executes (the largest block of `sharding.shard_bounds`), and adds a
MODELLED cost for every collective of the path:

    mask all-gather   bytes one rank receives from ONE peer / 153 GB/s (xGMI is point to point: the
                      N-1 peer blocks arrive over N-1 links in parallel) + a launch/sync latency
    small all-reduce  a fixed latency per call (joint models: the spectral sums, per EM iteration)

The latencies are ASSUMPTIONS (stated in the output), the per-rank times are measurements.  What
comes out is a model, not a scaling measurement: the driver's SCALE_rNN.json is the measurement.

BASELINE configs covered:
  configs[1]  one utterance F=513: bins sharded (the only axis there is)
  configs[2]  64 utterances through the chain: utterances sharded / bins sharded
  configs[4]  joint GCACGMM fit, bins sharded (spectral M-step sums all-reduced every iteration)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

XGMI_LINK_GBS = 153.0        # MI355X_MICROARCH.md: per xGMI link, per direction
GATHER_LATENCY_US = 20.0     # assumption: one RCCL all-gather call, launch + sync, small-message floor
ALLREDUCE_LATENCY_US = 35.0  # assumption: one RCCL all-reduce of a few KB over 8 ranks
F, T, D, K, E = 513, 500, 8, 3, 40
ITERS = 100
RANKS = (1, 2, 4, 8)


def med_ms(fn, reps, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(out))


def back_to_back_ms(fn, reps, warm=3):
    """Mean of `reps` launches issued back to back (what a rank in a loop sees; no drain per call)."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def main():
    import torch
    import bench_blocks as bench  # kernel_source_sha, make_batch
    from pb_bss_amd import _lib, engine, pipeline
    from pb_bss_amd.pipeline import device_ops as ops, _chain_after_masks
    from pb_bss_amd.sharding import shard_bounds
    from pb_bss_amd.testing import synth
    torch.cuda.set_device(0)
    out = {
        'what': 'MODEL, not a measurement of N GPUs: per-rank shares timed on one MI355X + modelled '
                'collectives (tools/scaling_model.py)',
        'assumptions': {
            'xgmi_link_GBps': XGMI_LINK_GBS, 'all_gather_latency_us': GATHER_LATENCY_US,
            'small_all_reduce_latency_us': ALLREDUCE_LATENCY_US,
            'note': 'direct all-gather over a fully connected xGMI mesh: a rank receives the N-1 '
                    'peer blocks over N-1 links in parallel, so the gather takes one block / link '
                    'bandwidth + latency; the slowest rank owns the largest block (65 of 513 bins)',
        },
        'kernel_source_sha': bench.kernel_source_sha(),
    }
    # ------------------------------------------------------------------ configs[1]
    Y0, init0 = synth.make_stft(F, T, D, K, seed=0)
    Y, init = _lib.to_device(Y0), _lib.to_device(init0)
    rows = []
    for n in RANKS:
        lo, hi = shard_bounds(F, n, 0)
        yb, ib = Y[lo:hi].contiguous(), init[lo:hi].contiguous()
        em = back_to_back_ms(lambda: ops.em_masks(yb, ib, ITERS), 30, 5)
        block = (hi - lo) * K * T * 8
        gather = 0.0 if n == 1 else block / (XGMI_LINK_GBS * 1e9) * 1e3 + GATHER_LATENCY_US * 1e-3
        rows.append({'ranks': n, 'bins_of_slowest_rank': hi - lo, 'em_ms_measured': em,
                     'gather_ms_modelled': gather, 'step_ms_model': em + gather})
    for r in rows:
        r['speedup_model'] = rows[0]['step_ms_model'] / r['step_ms_model']
    out['configs1_single_utterance_bins_sharded'] = {
        'rows': rows,
        'reading': 'one workgroup owns one bin for the whole EM loop: below 2 bins per compute unit '
                   'the launch time is the time of ONE bin\'s loop (100 iterations x ~12 us), '
                   'whatever the bin count -- a single utterance does not scale by sharding its bins',
    }
    # ------------------------------------------------------------------ configs[2]
    U = 64
    data = bench.make_batch(U)
    torch.cuda.empty_cache()
    Yb = _lib.to_device(np.stack([d[0] for d in data]))
    Ib = _lib.to_device(np.stack([d[1] for d in data]))
    utt, bins = [], []
    for n in RANKS:
        ulo, uhi = shard_bounds(U, n, 0)
        yu, iu = Yb[ulo:uhi].contiguous(), Ib[ulo:uhi].contiguous()
        step = back_to_back_ms(lambda: pipeline.separate(yu, iu, ITERS, 2 * (F - 1)), 5, 2)
        utt.append({'ranks': n, 'utterances_of_slowest_rank': uhi - ulo, 'step_ms_measured': step,
                    'collectives': 'none', 'step_ms_model': step})
        lo, hi = shard_bounds(F, n, 0)
        yl, il = Yb[:, lo:hi].contiguous(), Ib[:, lo:hi].contiguous()
        em = med_ms(lambda: ops.em_masks(yl, il, ITERS), 5)
        masks_all = ops.em_masks(Yb, Ib, ITERS) if n == 1 else masks_all  # noqa: F821
        mk = masks_all[ulo:uhi].transpose(-3, -2).contiguous()
        dhtv = med_ms(lambda: ops.dhtv_mapping(mk, 2 * (F - 1)), 5)
        mapping = ops.dhtv_mapping(masks_all.transpose(-3, -2).contiguous(), 2 * (F - 1)) \
            if n == 1 else mapping  # noqa: F821
        ml, mp = masks_all[:, lo:hi].contiguous(), mapping[..., lo:hi].contiguous()
        ext = med_ms(lambda: _chain_after_masks(yl, ml, mp, ops, 'gev+ban'), 5)
        block = U * (hi - lo) * K * T * 8
        gather = 0.0 if n == 1 else block / (XGMI_LINK_GBS * 1e9) * 1e3 + GATHER_LATENCY_US * 1e-3
        mgather = 0.0 if n == 1 else GATHER_LATENCY_US * 1e-3
        bins.append({'ranks': n, 'bins_of_slowest_rank': hi - lo, 'em_ms_measured': em,
                     'mask_gather_ms_modelled': gather, 'dhtv_ms_measured': dhtv,
                     'mapping_gather_ms_modelled': mgather, 'extract_ms_measured': ext,
                     'step_ms_model': em + gather + dhtv + mgather + ext,
                     'note': 'stages synchronised one after the other (they overlap in the real step)'})
    for rows_ in (utt, bins):
        for r in rows_:
            r['speedup_model'] = rows_[0]['step_ms_model'] / r['step_ms_model']
    out['configs2_batch64_chain'] = {
        'utterances_sharded': utt, 'bins_sharded': bins,
        'reading': 'utterance sharding has no collective and keeps every GPU at three workgroups '
                   'per compute unit down to 8 utterances per rank (4 104 bins): the share times ARE '
                   'the model; pipeline.separate picks it whenever there are at least as many '
                   'utterances as ranks',
    }
    del Yb, Ib, masks_all
    torch.cuda.empty_cache()
    # ------------------------------------------------------------------ configs[4]
    Yj, ej, ij = synth.make_joint(F, T, D, K, E, seed=0)
    yj, ee, gj = _lib.to_device(Yj), _lib.to_device(ej), _lib.to_device(ij)
    rows = []
    for n in RANKS:
        lo, hi = shard_bounds(F, n, 0)
        a, b, c = yj[lo:hi].contiguous(), ee[lo:hi].contiguous(), gj[lo:hi].contiguous()
        fit = back_to_back_ms(lambda: engine.joint_fit(a, b, K, _lib.EMBED_GAUSS_SPHERICAL, gamma0=c,
                                                       iterations=ITERS, final_predict=True,
                                                       check_status=False), 10, 3)
        ar = 0.0 if n == 1 else ITERS * ALLREDUCE_LATENCY_US * 1e-3
        block = (hi - lo) * K * T * 8
        gather = 0.0 if n == 1 else block / (XGMI_LINK_GBS * 1e9) * 1e3 + GATHER_LATENCY_US * 1e-3
        rows.append({'ranks': n, 'bins_of_slowest_rank': hi - lo, 'fit_ms_measured': fit,
                     'allreduce_ms_modelled': ar, 'gather_ms_modelled': gather,
                     'step_ms_model': fit + ar + gather})
    for r in rows:
        r['speedup_model'] = rows[0]['step_ms_model'] / r['step_ms_model']
    out['configs4_joint_bins_sharded'] = {
        'rows': rows,
        'reading': 'the sweep over the embedding shrinks with the rank\'s share of the points, the '
                   'spatial kernel (one workgroup per bin, serial per bin) does not, and every EM '
                   'iteration adds one small all-reduce of the spectral sums between two kernels',
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
