#!/usr/bin/env python3
"""Headline benchmark: cACGMM EM iterations/s on F=513, T=500, D=8, K=3
(BASELINE.json metric / configs[1]), one rank per GPU.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path: one fused `fit` of --iters (default 100)
EM iterations followed by the final E-step (fit_predict), with the complex64
observation already resident in HBM.  At N GPUs the job is N utterances
(weak scaling): every rank owns a contiguous block of ~513/N frequency bins of
EVERY utterance, runs the EM with no collective in the loop, and the posterior
masks are all-gathered over RCCL/xGMI at the end of each step (inside the timed region) (what
permutation alignment needs).  value = N * iters * K / max-over-ranks time.

The JSON line also carries
  roofline     -- algorithmic HBM bytes (8*F*T*D per EM iteration, SURVEY.md
                  section 8d) over the EM kernel's duration measured with HIP
                  events on the launch stream inside the library; plus the
                  FP64-VALU fraction, which is what actually binds the kernel;
  cpu_baseline -- the NumPy oracle (same einsum calls as the reference) timed
                  on this host on a bounded sample (rank 0, N = 1 only);
  mask_max_abs_err -- device vs oracle after all iterations on a bin subset.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F, T, D, K = 513, 500, 8, 3
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6  # MI355X FP64 vector peak
# float64 flops per frame per EM iteration (DESIGN.md "work model"):
#   outer product P (E phase) 192 + (M phase) 192, q_k 2*D*D*K, acc 2*D*D*K, softmax ~100
FLOPS_PER_FRAME_ITER = 2 * 192 + 2 * (2 * D * D * K) + 100


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=30)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--iters', type=int, default=100, help='EM iterations per fit')
    p.add_argument('--cpu-iters', type=int, default=100,
                   help='EM iterations of the CPU baseline sample (0 = skip)')
    p.add_argument('--check-bins', type=int, default=24,
                   help='bins of utterance 0 checked against the oracle (0 = skip)')
    return p.parse_args()


def pmc_traffic():
    """HBM bytes per launch (main + concurrent split kernel) from the committed rocprofv3 PMC
    passes of this same command (tools/profile_round.sh -> profiles/rNN_x_profile.txt):
    (FETCH_SIZE + WRITE_SIZE) KiB.  PMC passes cannot run inside the timed bench, so the
    figure is read back from the newest committed summary; None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                          'profiles', 'r*_profile.txt')))
    if not files:
        return None, None
    kib = 0.0
    for line in open(files[-1]):
        parts = [x.strip() for x in line.split('|')]
        if len(parts) == 4 and parts[1] in ('FETCH_SIZE', 'WRITE_SIZE'):
            kib += float(parts[3])
    if kib == 0.0:
        return None, None
    return kib * 1024.0, ('profiles/' + os.path.basename(files[-1]) +
                          ': FETCH_SIZE + WRITE_SIZE of both kernels, separate --pmc passes; '
                          '8-byte-per-lane loads, FETCH_SIZE not rescaled')


def pmc_sustained_clock_ghz():
    """Shader clock the EM kernel actually ran at in the committed profile: the persistent
    kernel keeps its waves resident for the whole launch, so SQ_WAVE_CYCLES (4-cycle units)
    per wave over the kernel duration is the clock.  None without a profile."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                          'profiles', 'r*_profile.txt')))
    if not files:
        return None
    cyc = waves = us = None
    for line in open(files[-1]):
        parts = [x.strip() for x in line.split('|')]
        if len(parts) == 4 and parts[0] == 'main' and parts[1] == 'SQ_WAVE_CYCLES':
            cyc = float(parts[3])
        if len(parts) == 4 and parts[0] == 'main' and parts[1] == 'SQ_WAVES':
            waves = float(parts[3])
        if len(parts) == 6 and 'cacgmm_em_kernel' in parts[0]:
            us = float(parts[2])
    if not (cyc and waves and us):
        return None
    return cyc * 4.0 / waves / (us * 1e3)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from pb_bss_amd.testing import synth  # input generator shared with the parity tests
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.sharding import all_gather_bins, shard_bounds

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # under torch.distributed.run (RANK set) always go through RCCL, also for 1 rank
    use_dist = 'RANK' in os.environ
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', device_id=dev)

    # ---- workload: `world` utterances, this rank's block of bins of each ----
    lo, hi = shard_bounds(F, world, rank)
    ys, inits = [], []
    Y0 = init0 = None
    for u in range(world):
        Y, init = synth.make_stft(F, T, D, K, seed=u)
        if u == 0:
            Y0, init0 = Y, init
        ys.append(Y[lo:hi])
        inits.append(init[lo:hi])
    y = _lib.to_device(np.concatenate(ys))        # (world*(hi-lo), T, D) complex64
    g0 = _lib.to_device(np.concatenate(inits))    # (world*(hi-lo), K, T) float64
    n_loc = hi - lo
    engine.set_timing(True, local_rank)

    def step():
        r = engine.em_fit(y, K, gamma0=g0, iterations=args.iters, final_predict=True,
                          check_status=False)
        ms = engine.last_kernel_ms(local_rank)  # HIP events on the launch stream
        masks = r['affiliation'].reshape(world, n_loc, K, T)
        if use_dist:
            # the one exchange step of the path: all-gather the masks over RCCL/xGMI, in stream
            # order after the EM kernel.  (Overlapping it with the NEXT step's EM on a side
            # stream was measured to be slower: whatever occupies a CU while the persistent
            # kernel's workgroups are being placed skews their distribution for the whole
            # launch -- 1.71 -> 2.45 ms per EM kernel with one rank.)
            masks = all_gather_bins(masks, F, bin_axis=1)
        return masks, ms, r

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        masks, ms, r = step()
        kernel_ms += ms
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        value = world * args.iters * args.steps / elapsed
        avg_kernel_s = kernel_ms / args.steps * 1e-3
        alg_bytes = 8.0 * (world * n_loc) * T * D * args.iters  # per launch, this rank
        achieved = alg_bytes / avg_kernel_s / 1e9
        flops = FLOPS_PER_FRAME_ITER * (world * n_loc) * T * args.iters
        tflops = flops / avg_kernel_s / 1e12
        traffic, traffic_src = (pmc_traffic() if (world == 1 and args.iters == 100)
                                else (None, None))
        clock_ghz = pmc_sustained_clock_ghz()
        out = {
            'metric': 'cACGMM EM iterations/sec on F=513,T=500,D=8,K=3',
            'value': value,
            'unit': 'EM iterations/s (utterance-iterations, whole job)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[1]: 8-mic 3-source cACGMM, F=513 T=500 D=8 K=3, '
                            'complex64 STFT resident in HBM, fit_predict',
                'em_iterations_per_step': args.iters, 'utterances': world,
                'sharding': (f'frequency bins, {n_loc} of {F} per rank per utterance; RCCL mask '
                             'all-gather per step, in stream order after the EM kernel') if use_dist else 'none (1 GPU)',
            },
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'traffic_source': traffic_src,
                'kernel': 'cacgmm_em_kernel<8,3,float,false> (+ concurrent cacgmm_em_split_kernel for '
                          'the remainder bins; kernel_ms brackets both)',
                'kernel_ms': avg_kernel_s * 1e3,
                'algorithmic_bytes_per_launch': alg_bytes,
                'note': 'y stays in LDS for the whole EM loop; the binding resource is '
                        'FP64 VALU, see fp64_valu',
                'fp64_valu': {'achieved': tflops, 'peak': FP64_VALU_PEAK_TF, 'unit': 'TFLOP/s',
                              'frac': tflops / FP64_VALU_PEAK_TF,
                              'flops_per_frame_iter': FLOPS_PER_FRAME_ITER,
                              'sustained_clock_ghz': clock_ghz,
                              'frac_at_sustained_clock': (
                                  None if not clock_ghz else
                                  tflops / (FP64_VALU_PEAK_TF * clock_ghz / 2.4)),
                              'note': 'peak assumes 2.4 GHz; sustained_clock_ghz = SQ_WAVE_CYCLES '
                                      'per resident wave / kernel time in the committed profile'},
            },
        }
        st = _lib.to_host(r['status'])
        out['status_bits_or'] = int(np.bitwise_or.reduce(st.ravel()))
        # ---- parity on a bin subset of utterance 0 (oracle = checker only) ----
        if args.check_bins and lo == 0:
            from oracle import cacgmm as oc
            nb = min(args.check_bins, n_loc)
            Y128 = Y0[:nb].astype(np.complex128)
            m = oc.em_fit(Y128, init0[:nb], iterations=args.iters)
            ref = oc.em_predict(m, Y128)
            got = _lib.to_host(masks[0, :nb])
            out['mask_max_abs_err'] = float(np.abs(got - ref).max())
            out['mask_err_bins_checked'] = nb
        # ---- CPU baseline: NumPy oracle on this host, bounded sample ----------
        if world == 1 and args.cpu_iters > 0:
            from oracle import cacgmm as oc
            Y128 = Y0.astype(np.complex128)
            t1 = time.perf_counter()
            oc.em_fit(Y128, init0, iterations=args.cpu_iters)
            dt = time.perf_counter() - t1
            out['cpu_baseline'] = {
                'value': args.cpu_iters / dt, 'unit': 'EM iterations/s', 'cores': 1,
                'kind': 'port',
                'sample': f'NumPy oracle (reference einsum calls, float64), full F=513 T=500 '
                          f'D=8 K=3, {args.cpu_iters} EM iterations, {dt:.1f} s; '
                          f'host has {os.cpu_count()} logical cores, einsum is single-threaded',
            }
        line = json.dumps(out)
    else:
        line = None
    if use_dist:
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes a version banner through C stdio; when stdout is a pipe it sits in libc's
        # buffer until exit and would land AFTER the JSON line: flush it out first, so that the
        # result is the last line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == '__main__':
    main()
