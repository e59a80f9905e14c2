#!/usr/bin/env python3
"""End-to-end mask-based separation of a batch of utterances on one or more GPUs
(the shape of BASELINE config 3):

    STFT batch (U, F, T, D)  --cACGMM EM (frequency bins sharded over ranks)-->  masks (U, F, K, T)
      --RCCL all-gather-->  DHTV permutation alignment  -->  PSD  -->  'gev+ban' beamformer  -->  (U, K, F, T)

Single GPU:   python examples/separate_batch.py --utterances 4
N GPUs:       python -m torch.distributed.run --nproc-per-node N examples/separate_batch.py --utterances 8

Everything between the input STFT and the enhanced STFT stays on the device(s).
Synthetic input (pb_bss_amd/testing/synth.py generator, shared with the tests).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def separate_by_utterance(Y, init, iterations, stft_size, group=None):
    """Utterance-level data parallelism (preferable when U >= ranks): every rank
    runs the whole chain on its own block of utterances with NO collective in between and the
    results are all-gathered once at the end."""
    import torch.distributed as dist
    from pb_bss_amd.sharding import all_gather_bins, shard_bounds
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    U = Y.shape[0]
    assert U >= world, (U, world)
    lo, hi = shard_bounds(U, world, rank)
    out = separate(Y[lo:hi], init[lo:hi], iterations, stft_size, group=None, sharded=False)
    return {k: all_gather_bins(v.contiguous(), U, bin_axis=0, group=group) for k, v in out.items()}


def separate(Y, init, iterations, stft_size, group=None, sharded=True):
    """Y (U, F, T, D) complex torch-CUDA or NumPy; init (U, F, K, T).
    Returns dict(masks (U, K, F, T) aligned, enhanced (U, K, F, T), mapping)."""
    import torch
    import torch.distributed as dist
    from pb_bss_amd import _lib
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.extraction import (apply_beamforming_vector, get_bf_vector,
                                       get_power_spectral_density_matrix)
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from pb_bss_amd.sharding import fit_predict_sharded

    Y = _lib.to_device(Y)
    init = _lib.to_device(init, torch.float64)
    if sharded and dist.is_available() and dist.is_initialized():
        masks = fit_predict_sharded(Y, init, iterations=iterations, bin_axis=-3, group=group)
    else:
        masks = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=iterations)
    kft = masks.transpose(-3, -2).contiguous()                      # (U, K, F, T)
    solver = DHTVPermutationAlignment.from_stft_size(stft_size)
    mapping = solver.calculate_mapping(kft)
    aligned = solver.apply_mapping(kft, mapping)                    # (U, K, F, T)
    X = Y.transpose(-2, -1).contiguous()                            # (U, F, D, T)
    psd = get_power_spectral_density_matrix(X, aligned.transpose(-3, -2).contiguous())  # (U,F,K,D,D)
    K = psd.shape[-3]
    enhanced = []
    for k in range(K):
        target = psd[..., k, :, :]
        noise = psd.sum(dim=-3) - target
        w = get_bf_vector('gev+ban', target, noise)                 # (U, F, D)
        enhanced.append(apply_beamforming_vector(w, X))             # (U, F, T)
    return dict(masks=aligned, enhanced=torch.stack(enhanced, dim=1),
                mapping=_lib.to_device(mapping) if not _lib.is_torch(mapping) else mapping)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--utterances', type=int, default=4)
    ap.add_argument('--iterations', type=int, default=100)
    ap.add_argument('--F', type=int, default=513)
    ap.add_argument('--T', type=int, default=500)
    ap.add_argument('--D', type=int, default=8)
    ap.add_argument('--K', type=int, default=3)
    ap.add_argument('--shard', choices=['bins', 'utterances'], default='bins',
                    help='multi-GPU: frequency bins of every utterance over the ranks (one mask '
                         'all-gather), or whole utterances per rank (no collective until the end)')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from pb_bss_amd.testing import synth  # synthetic input generator
    use_dist = 'RANK' in os.environ
    if use_dist:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')
    rank = dist.get_rank() if use_dist else 0
    stft_size = 2 * (args.F - 1)
    data = [synth.make_stft(args.F, args.T, args.D, args.K, seed=s) for s in range(args.utterances)]
    Y = np.stack([d[0] for d in data])
    init = np.stack([d[1] for d in data])
    from pb_bss_amd import _lib
    Yd, initd = _lib.to_device(Y), _lib.to_device(init)
    separate(Yd[:1], initd[:1], 2, stft_size)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if use_dist and args.shard == 'utterances':
        out = separate_by_utterance(Yd, initd, args.iterations, stft_size)
    else:
        out = separate(Yd, initd, args.iterations, stft_size)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f'{args.utterances} utterances x {args.iterations} EM iterations + DHTV + gev+ban: '
              f'{dt * 1e3:.1f} ms ({dt / args.utterances * 1e3:.2f} ms/utterance); '
              f'masks {tuple(out["masks"].shape)}, enhanced {tuple(out["enhanced"].shape)}')
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
