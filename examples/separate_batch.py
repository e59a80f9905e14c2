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


def separate(Y, init, iterations, stft_size, group=None, sharded=True, shard='bins'):
    """Y (U, F, T, D) complex torch-CUDA or NumPy; init (U, F, K, T).  Thin wrapper over
    `pb_bss_amd.pipeline.separate` (the chain itself lives in the package so that bench.py,
    this example and the tests run the same code); gathers the outputs on every rank."""
    import torch
    import torch.distributed as dist
    from pb_bss_amd import _lib, pipeline
    Y = _lib.to_device(Y)
    init = _lib.to_device(init, torch.float64)
    multi = sharded and dist.is_available() and dist.is_initialized()
    return pipeline.separate(Y, init, iterations, stft_size, shard=shard if multi else None,
                             group=group, gather_output=True)


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
    separate(Yd[:1], initd[:1], 2, stft_size, sharded=False)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = separate(Yd, initd, args.iterations, stft_size, shard=args.shard)
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
