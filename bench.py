#!/usr/bin/env python3
"""Headline benchmark: cACGMM EM iterations/s on F=513, T=500, D=8, K=3
(BASELINE.json metric / configs[1]), one rank per GPU.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path: one fused `fit` of --iters (default 100) EM iterations
followed by the final E-step (fit_predict), with the complex64 observation already resident in
HBM.  At N GPUs the job is N utterances (weak scaling): every rank owns a contiguous block of
~513/N frequency bins of EVERY utterance, runs the EM with no collective in the loop, and the
posterior masks are all-gathered over RCCL/xGMI at the end of each step (inside the timed region)
-- what permutation alignment needs.  value = N * iters * K / max-over-ranks time.

This file holds the timing contract (fence / timed / max over ranks), the headline workload and
the emitter.  The further workloads -- `config3` (BASELINE configs[2]: 64 utterances through EM ->
DHTV -> PSD -> gev+ban -> apply, the STRONG-scaling workload), `config4` (configs[3], Watson / vMF
+ MVDR-Souden), `config5` (configs[4], joint GCACGMM), `f32` (packed-FP32 instantiation), the
secondary single-utterance figures and the CPU baselines -- live in tools/bench_blocks.py.

Output: the LAST line on stdout is one strict-JSON object of at most 8 KB (`compact_line`): the
contract keys, `roofline`, `cpu_baseline`, `verify` and one-number summaries of every other
workload.  The full blocks go to `bench_extra.json` (in gpurun_out/ when that directory exists,
else the working directory; `--extra-file` overrides).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'tools')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

F, T, D, K = 513, 500, 8, 3
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6  # MI355X FP64 vector peak
FP32_VALU_PEAK_TF = 157.3  # MI355X FP32 vector peak (packed)
# float64 flops per frame per EM iteration: Hermitian outer product P in the E
# phase (192) and again in the M phase (192; it cannot stay resident: 256 KB per bin),
# q_k = <A_k, P> (2 D^2 K), C_k += w_k P (2 D^2 K), softmax (~100)
FLOPS_PER_FRAME_ITER = 2 * 192 + 2 * (2 * D * D * K) + 100
LINE_LIMIT = 8192         # bytes of the final stdout line (the driver's parser lost a 27 KB line)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=30)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--iters', type=int, default=100, help='EM iterations per fit')
    p.add_argument('--cpu-iters', type=int, default=100,
                   help='EM iterations of the CPU baseline sample (0 = skip)')
    p.add_argument('--extras', choices=('auto', 'off'), default='auto',
                   help="'off': skip the secondary single-utterance figures (profiling runs want "
                        "the timed steps to be the last EM launches of the process)")
    p.add_argument('--check-bins', type=int, default=24,
                   help='bins checked against the oracle per utterance at N = 1; at N > 1 '
                        'max(3, check_bins / N^2) bins of EVERY rank\'s shard of EVERY utterance '
                        '(0 = skip)')
    p.add_argument('--preheat-s', type=float, default=1.0,
                   help='seconds of untimed back-to-back steps BEFORE the W warm-up steps: from idle '
                        'the chip needs tens of launches to ramp its shader clock up (kernel trace: '
                        '1.83 -> 1.61 ms over the first 30 launches); 0 = off')
    p.add_argument('--sustained-s', type=float, default=2.0,
                   help='seconds of back-to-back steps after the timed region (0 = skip)')
    p.add_argument('--workload', choices=['config2', 'config3', 'config4', 'config5'],
                   default='config2',
                   help='config2: BASELINE configs[1] (headline, with summaries of the config-3/4/5 '
                        'blocks in the same line); config3 / config4 / config5: configs[2] / [3] / '
                        '[4] as the primary line (what tools/profile_workload.sh profiles)')
    p.add_argument('--configs45', choices=['auto', 'off'], default='auto',
                   help='auto: also run BASELINE configs[3] and configs[4] (N = 1 only)')
    p.add_argument('--c4-extraction', choices=['on', 'off'], default='on',
                   help='--workload config4: off = the step is the mixture fit alone')
    p.add_argument('--leg', choices=['watson', 'vmf'], default='watson',
                   help='--workload config4: which mixture is the primary line')
    p.add_argument('--config3', choices=['auto', 'off'], default='auto',
                   help='auto: also run BASELINE configs[2] (64 utterances through the chain) -- at '
                        'N > 1 with the bins and with the utterances sharded; off: headline only')
    p.add_argument('--config3-steps', type=int, default=5)
    p.add_argument('--shard', choices=['bins', 'utterances'], default='bins',
                   help='--workload config3 with N > 1: what is sharded over the ranks')
    p.add_argument('--utterances', type=int, default=64, help='config3 batch size')
    p.add_argument('--mask-gather', choices=['f64', 'f32'], default='f64',
                   help='config3, bins sharded: dtype of the mask all-gather')
    p.add_argument('--beamformer', choices=['gev+ban', 'mvdr_souden'], default='gev+ban',
                   help='config3 extraction stage')
    p.add_argument('--comm', choices=['torch', 'native'], default='torch',
                   help='mask all-gather through torch.distributed (RCCL) or through the '
                        'library\'s own RCCL communicator (C ABI pbbss_allgather_masks)')
    p.add_argument('--f32', choices=['auto', 'off'], default='auto',
                   help='auto: also measure the packed-FP32 instantiation when the library has it')
    p.add_argument('--extra-file', default=None,
                   help='where the full blocks go (default gpurun_out/bench_extra.json when '
                        'gpurun_out/ exists, else ./bench_extra.json; "none" = nowhere)')
    p.add_argument('--print-source-sha', action='store_true',
                   help='print the hash of the kernel sources of --workload (tools/profile_round.sh '
                        'stamps it into the profile summaries) and exit')
    return p.parse_args()


# ------------------------------------------------------------------------------------------
# the timing contract
# ------------------------------------------------------------------------------------------
def setup(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, (world, args.gpus)
    # PBBSS_BENCH_ONE_DEVICE=1 (development only): every rank on GPU 0 with the gloo backend -- a
    # FUNCTIONAL rehearsal of the N > 1 control flow on a one-GPU box, where RCCL refuses two ranks
    # on one device; the figures of such a run mean nothing and the line says so (`rehearsal`)
    one_dev = os.environ.get('PBBSS_BENCH_ONE_DEVICE') == '1'
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = 'RANK' in os.environ  # under torch.distributed.run always go through RCCL
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if one_dev:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
        if args.comm == 'native':
            from pb_bss_amd import sharding
            sharding.init_native_comm(device_index=local_rank)
    return world, rank, local_rank, dev, use_dist


def fence(use_dist):
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, use_dist, dev):
    import torch
    import torch.distributed as dist
    if not use_dist:
        return x
    tt = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def preheat(step, seconds, use_dist, dev):
    """Untimed steps until `seconds` have passed on the slowest rank (every rank runs the same
    count: the step may contain a collective).  -> {'seconds', 'steps'} or None."""
    if seconds <= 0:
        return None
    fence(use_dist)
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    fence(use_dist)
    per = max_over_ranks((time.perf_counter() - t0) / 5, use_dist, dev)
    n = max(0, int(math.ceil(seconds / max(per, 1e-6))) - 5)
    for _ in range(n):
        step()
    fence(use_dist)
    return {'seconds': time.perf_counter() - t0, 'steps': n + 5,
            'note': 'untimed clock ramp before the W warm-up steps (--preheat-s)'}


def run_steps(step, n, read_ms):
    """n steps; with read_ms the kernel time of EVERY one of them is collected -- two launches
    late inside the loop, the last two after it -- without ever draining the queue.  -> (last
    result, summed kernel ms)."""
    last, kms = None, 0.0
    for i in range(n):
        last = step()
        if read_ms is not None and i >= 2:
            kms += read_ms(2)
    if read_ms is not None:
        for lag in range(min(n, 2) - 1, -1, -1):
            kms += read_ms(lag)
    return last, kms


def timed(step, steps, warmup, use_dist, dev, read_ms=None):
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize; max over ranks."""
    for _ in range(warmup):
        step()
    fence(use_dist)
    t0 = time.perf_counter()
    last, kernel_ms = run_steps(step, steps, read_ms)
    fence(use_dist)
    elapsed = max_over_ranks(time.perf_counter() - t0, use_dist, dev)
    return elapsed, kernel_ms / max(steps, 1), last


def sustained(step, seconds, ms_per_step, min_steps, use_dist, dev, read_ms=None):
    """The same step back to back for >= `seconds` after the timed region (DVFS give-back)."""
    if seconds <= 0:
        return None
    n = max(min_steps, int(math.ceil(seconds * 1e3 / max(ms_per_step, 1e-3))))
    fence(use_dist)
    t0 = time.perf_counter()
    _, kms = run_steps(step, n, read_ms)
    fence(use_dist)
    el = max_over_ranks(time.perf_counter() - t0, use_dist, dev)
    return {'seconds': el, 'steps': n, 'ms_per_step': el / n * 1e3, 'kernel_ms': kms / n,
            'note': 'run right after the timed steps; a higher ms_per_step than the timed region '
                    'is the chip\'s power management giving clock back under a sustained load'}


def roofline_block(kernel_ms, bins, iters, ms_per_step=None, world_note='', peak_tf=FP64_VALU_PEAK_TF,
                   bound='fp64_valu', kernel='cacgmm_em_kernel<8,3,float,false>'):
    """roofline object of the EM kernel.  `achieved` = useful flops of one launch / kernel_ms, the
    kernel's average duration measured LIVE with HIP events on its dispatch (library, launch
    stream).  float64 VALU is what bounds it -- the observation is LDS-resident, HBM sees y once
    per fit -- so the section-8d HBM contract figure (8*F*T*D bytes per EM iteration against
    8 TB/s) rides along as `hbm_contract`: at 1 252 flops per frame-iteration, 50 % of that figure
    IS 100 % of the FP64 vector peak.  `traffic` / `clock_ghz` come from separate --pmc passes:
    a committed profile of exactly these kernel sources (tools/bench_blocks.pmc_evidence)."""
    import bench_blocks as bb
    avg_kernel_s = kernel_ms * 1e-3
    alg_bytes = 8.0 * bins * T * D * iters
    hbm = alg_bytes / avg_kernel_s / 1e9
    tflops = FLOPS_PER_FRAME_ITER * bins * T * iters / avg_kernel_s / 1e12
    if bins == F and iters == 100 and bound == 'fp64_valu':
        traffic, clock, traffic_src = bb.pmc_evidence(ms_per_step)
    else:
        traffic, clock, traffic_src = None, None, \
            'PMC traffic is only collected for the single-utterance headline command'
    return {
        'bound': bound, 'achieved': tflops, 'peak': peak_tf, 'unit': 'TFLOP/s',
        'frac': tflops / peak_tf, 'traffic': traffic, 'traffic_source': traffic_src,
        'kernel': kernel + ' (kernel_ms: HIP events on its dispatch, average over the timed steps; '
                           'the remainder bin runs as member workgroups)' + world_note,
        'kernel_ms': kernel_ms,
        'flops_per_frame_iter': FLOPS_PER_FRAME_ITER,
        'clock_ghz': clock,
        'frac_at_sustained_clock': None if not clock else tflops / (peak_tf * clock / 2.4),
        'hbm_contract': {
            'bound': 'hbm', 'achieved': hbm, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': hbm / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': alg_bytes,
            'note': 'SURVEY.md 8d: 8*F*T*D bytes per EM iteration against 8 TB/s; y stays in LDS '
                    'for the whole EM loop, HBM does not bind',
        },
    }


# ------------------------------------------------------------------------------------------
# headline: BASELINE configs[1]
# ------------------------------------------------------------------------------------------
def run_headline(args, world, rank, local_rank, dev, use_dist, precision='f64'):
    import torch
    import bench_blocks as bb
    from pb_bss_amd.testing import synth  # input generator shared with the parity tests
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.sharding import all_gather_bins, shard_bounds

    # ---- workload: `world` utterances, this rank's block of bins of each ----
    lo, hi = shard_bounds(F, world, rank)
    data = [synth.make_stft(F, T, D, K, seed=u) for u in range(world)]
    y = _lib.to_device(np.concatenate([d[0][lo:hi] for d in data]))   # (world*(hi-lo), T, D) c64
    g0 = _lib.to_device(np.concatenate([d[1][lo:hi] for d in data]))  # (world*(hi-lo), K, T) f64
    n_loc = hi - lo
    engine.set_timing(True, local_rank)
    fit_kw = {} if precision == 'f64' else {'precision': 'f32'}

    def step():
        r = engine.em_fit(y, K, gamma0=g0, iterations=args.iters, final_predict=True,
                          check_status=False, **fit_kw)
        masks = r['affiliation'].reshape(world, n_loc, K, T)
        if use_dist:
            # the one exchange step of the path: all-gather the masks over RCCL/xGMI, in stream
            # order after the EM kernel (overlapping it with the NEXT step's EM on a side stream
            # was measured slower: 1.71 -> 2.45 ms per EM kernel with one rank)
            masks = all_gather_bins(masks, F, bin_axis=1)
        return {'masks': masks, 'r': r}

    # HIP events on the dispatch of the EM kernel (library, launch stream), read TWO launches late:
    # reading the most recent launch would drain the queue every step
    read_ms = lambda lag: engine.last_kernel_ms(local_rank, lag)  # noqa: E731
    ph = preheat(step, args.preheat_s, use_dist, dev)
    elapsed, kernel_ms, last = timed(step, args.steps, args.warmup, use_dist, dev, read_ms)
    ms_per_step = elapsed / args.steps * 1e3
    sus = sustained(step, args.sustained_s, ms_per_step, args.steps, use_dist, dev, read_ms)
    masks, r = last['masks'], last['r']

    # ---- the exchange step on its own (untimed region): HIP events around the gather ----
    gather_ms = None
    mine = {'em_kernel_ms': kernel_ms, 'bins_per_utterance': float(n_loc)}
    if use_dist:
        loc = r['affiliation'].reshape(world, n_loc, K, T)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence(use_dist)
        e0.record()
        for _ in range(10):
            all_gather_bins(loc, F, bin_axis=1)
        e1.record()
        torch.cuda.synchronize()
        mine['gather_ms'] = e0.elapsed_time(e1) / 10
        gather_ms = max_over_ranks(mine['gather_ms'], use_dist, dev)
    same = bb.identical_on_all_ranks(masks, use_dist, world)
    per_rank = bb.per_rank_table(mine, use_dist, world, dev)
    got2 = None
    if precision != 'f64' and args.check_bins:
        # per-step check of the single-precision kernel (see the verification below): two
        # iterations from the same initialisation, a separate untimed launch on every rank
        r2 = engine.em_fit(y, K, gamma0=g0, iterations=2, final_predict=True, check_status=False,
                           **fit_kw)
        m2 = r2['affiliation'].reshape(world, n_loc, K, T)
        if use_dist:
            m2 = all_gather_bins(m2, F, bin_axis=1)
        got2 = _lib.to_host(m2)
    st = _lib.to_host(r['status'])
    status_or = int(np.bitwise_or.reduce(st.ravel()))
    if use_dist:  # NCCL has no bitwise-or reduction: gather the words, OR them here
        import torch.distributed as dist
        tt = torch.tensor([status_or], dtype=torch.int64, device=dev)
        allst = torch.empty((world,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allst, tt)
        status_or = int(np.bitwise_or.reduce(_lib.to_host(allst)))
    if rank != 0:
        return None

    out = {
        'metric': 'cACGMM EM iterations/sec on F=513,T=500,D=8,K=3',
        'value': world * args.iters * args.steps / elapsed,
        'unit': 'EM iterations/s (utterance-iterations, whole job)',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'scaling_note': 'WEAK by construction: N utterances on N GPUs (~513 bins per GPU whatever '
                        'N).  The STRONG curve is `strong` (BASELINE configs[2]: a fixed batch of '
                        '64 utterances through the whole chain).',
        'dtype': 'f64' if precision == 'f64' else 'f32', 'data': 'synthetic',
        'config': {
            'workload': 'BASELINE configs[1]: 8-mic 3-source cACGMM, F=513 T=500 D=8 K=3, '
                        'complex64 STFT resident in HBM, fit_predict' +
                        ('' if precision == 'f64' else
                         ' -- REFERENCE-PRECISION instantiation: packed-FP32 E/M phases (the '
                         'reference\'s own arithmetic for complex64 input + ndarray '
                         'initialisation, cacgmm.py:226-227), float64 class sums and '
                         'factorisation; NOT the headline'),
            'em_iterations_per_step': args.iters, 'utterances': world,
            'sharding': (f'frequency bins, {n_loc} of {F} per rank per utterance; RCCL mask '
                         'all-gather per step (' + args.comm + ' communicator), in stream order '
                         'after the EM kernel') if use_dist else 'none (1 GPU)',
        },
        'status_bits_or': status_or,
        'preheat': ph,
        'sustained': sus,
        'per_rank': dict(per_rank, what='per rank: EM kernel ms per step (HIP events in the library, '
                                        'average over the timed steps), its bins per utterance, '
                                        'and -- N > 1 -- its time for one mask all-gather'),
    }
    if precision == 'f64':
        out['roofline'] = roofline_block(kernel_ms, world * n_loc, args.iters, ms_per_step)
    else:
        out['roofline'] = roofline_block(
            kernel_ms, world * n_loc, args.iters, ms_per_step, peak_tf=FP32_VALU_PEAK_TF,
            bound='fp32_valu', kernel='cacgmm_em32_kernel<8,3>')
    if sus:
        out['sustained']['value'] = world * args.iters * sus['steps'] / sus['seconds']
    cb = bb.comm_block(args, use_dist, world, (world * F * K * T * 8) * (world - 1) // max(world, 1),
                       gather_ms, f'all-gather of the posterior masks ({world} utterances x {F} bins '
                                  f'x {K} x {T} float64) once per step')
    if cb:
        out['comm'] = cb
    # ---- parity: bins of EVERY rank's shard of EVERY utterance (oracle = checker only) ----
    if args.check_bins:
        bb.verify_headline(out, args, world, data, _lib.to_host(masks), got2, same, precision)
    return out, data[0]


# ------------------------------------------------------------------------------------------
# the emitter: one strict-JSON line of at most LINE_LIMIT bytes, the full blocks to a side file
# ------------------------------------------------------------------------------------------
def compact_line(full, extra_path=None):
    """The driver's line: contract keys + roofline + cpu_baseline + verify in full (long notes
    cut), every other workload as a one-number summary; strict JSON, <= LINE_LIMIT bytes."""
    from bench_blocks import strict, _sig, _summary, strong_block
    full = strict(full)
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
            'higher_is_better', 'scaling', 'scaling_note', 'vs_baseline', 'dtype', 'data', 'config')
    out = {k: full[k] for k in keep if k in full}
    rf = dict(full.get('roofline') or {})
    if rf:
        if isinstance(rf.get('hbm_contract'), dict):
            rf['hbm_contract'] = {k: rf['hbm_contract'][k] for k in
                                  ('bound', 'achieved', 'peak', 'unit', 'frac',
                                   'algorithmic_bytes_per_launch') if k in rf['hbm_contract']}
        for k in ('traffic_source', 'kernel'):
            if isinstance(rf.get(k), str) and len(rf[k]) > 200:
                rf[k] = rf[k][:197] + '...'
        out['roofline'] = rf
    cb = full.get('cpu_baseline')
    if isinstance(cb, dict):
        out['cpu_baseline'] = {k: (v[:297] + '...' if isinstance(v, str) and len(v) > 300 else v)
                               for k, v in cb.items()
                               if k in ('value', 'unit', 'cores', 'kind', 'sample', 'runs',
                                        'reference_recorded')}
    v = full.get('verify')
    if isinstance(v, dict):
        out['verify'] = {k: v[k] for k in v if k != 'what'}
        out['mask_max_abs_err'] = full.get('mask_max_abs_err')
    for k in ('status_bits_or', 'rehearsal'):
        if k in full:
            out[k] = full[k]
    if isinstance(full.get('sustained'), dict):
        out['sustained'] = {k: full['sustained'].get(k) for k in ('value', 'ms_per_step', 'steps')}
    if isinstance(full.get('per_rank'), dict):
        out['per_rank'] = {k: v for k, v in full['per_rank'].items() if k != 'what'}
    if isinstance(full.get('comm'), dict):
        c = full['comm']
        out['comm'] = {'backend': c.get('backend'), 'world_size': c.get('world_size'),
                       'communicator': c.get('communicator'),
                       'ranks_seen_by_rccl': c.get('rccl_comm_count', c.get('world_size')),
                       'gather_ms_max': c.get('gather_ms'),
                       'bytes_per_rank': c.get('bytes_received_per_rank_per_step'),
                       'gather_GBps_per_rank': c.get('gather_GBps_per_rank')}
    st = strong_block(full.get('config3'), full.get('n_gpus', 1))
    if st:
        out['strong'] = st
    if 'f32' in full:
        out['f32'] = _summary(full['f32'])
    c3 = full.get('config3')
    if isinstance(c3, dict):
        out['config3'] = {k: _summary(v) for k, v in c3.items()
                          if isinstance(v, dict) and 'value' in v}
        if 'error' in c3:
            out['config3']['error'] = str(c3['error'])[:160]
        if 'mapping_identical_across_shardings' in c3:
            out['config3']['mapping_identical_across_shardings'] = \
                c3['mapping_identical_across_shardings']
    c4 = full.get('config4')
    if isinstance(c4, dict):
        out['config4'] = {'watson': _summary(c4), 'vmf': _summary(c4.get('vmf'))}
    if 'config5' in full:
        out['config5'] = _summary(full['config5'])
    for k, sub in (('host_numpy_in_out', ('ms_per_call', 'value')),
                   ('single_utterance_chain', ('em_fit_predict_ms', 'dhtv_mapping_ms',
                                               'align_psd_bf_apply_ms', 'total_ms'))):
        if isinstance(full.get(k), dict):
            out[k] = ({'error': str(full[k]['error'])[:160]} if 'error' in full[k] else
                      {s: _sig(full[k].get(s)) for s in sub})
    out['extra_file'] = extra_path
    out['extra_keys'] = sorted(k for k in full if k not in out)
    line = json.dumps(out, allow_nan=False, separators=(',', ':'))
    # belt and braces: whatever a future block adds, the line never outgrows the driver's parser
    for k in ('extra_keys', 'single_utterance_chain', 'host_numpy_in_out', 'per_rank', 'sustained',
              'config5', 'config4', 'config3', 'f32'):
        if len(line.encode()) <= LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, allow_nan=False, separators=(',', ':'))
    if len(line.encode()) > LINE_LIMIT:   # last resort: cut every long string

        def cut(o):
            if isinstance(o, dict):
                return {k: cut(v) for k, v in o.items()}
            if isinstance(o, list):
                return [cut(v) for v in o[:16]]
            return o[:117] + '...' if isinstance(o, str) and len(o) > 120 else o
        line = json.dumps(cut(out), allow_nan=False, separators=(',', ':'))
    assert len(line.encode()) <= LINE_LIMIT, len(line)
    return line


def extra_path_for(args):
    if args.extra_file == 'none':
        return None
    if args.extra_file:
        return args.extra_file
    d = os.path.join(os.getcwd(), 'gpurun_out')
    return os.path.join(d if os.path.isdir(d) and os.access(d, os.W_OK) else os.getcwd(),
                        'bench_extra.json')


def write_extra(full, path):
    """The full blocks -> `path` (strict JSON, indented).  -> the path written, or None."""
    if path is None:
        return None
    try:
        with open(path, 'w') as f:
            from bench_blocks import strict
            json.dump(strict(full), f, allow_nan=False, indent=1)
        return os.path.relpath(path)
    except OSError as e:
        print(f'bench.py: could not write {path}: {e}', file=sys.stderr)
        return None


def emit(full, use_dist, args=None):
    """Tear the process group down, then print `full` (rank 0; None elsewhere) as the compact
    line -- the LAST line on stdout."""
    if use_dist:
        import torch.distributed as dist
        from pb_bss_amd import sharding
        sharding.destroy_native_comm()
        dist.destroy_process_group()
    if full is None:
        return
    path = write_extra(full, extra_path_for(args) if args is not None else None)
    line = compact_line(full, path)
    # RCCL writes a version banner through C stdio; when stdout is a pipe it sits in libc's buffer
    # until exit and would land AFTER the JSON line: flush it out first
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(line, flush=True)


def main():
    args = parse()
    import bench_blocks as bb
    if args.print_source_sha:
        if args.workload in ('config3', 'config4', 'config5'):
            print(bb.workload_source_sha('config4_vmf' if (args.workload, args.leg) ==
                                         ('config4', 'vmf') else args.workload))
        else:
            print(bb.kernel_source_sha())
        return
    if args.workload == 'config3':
        return bb.main_config3(args)
    if args.workload in ('config4', 'config5'):
        return bb.main_config45(args)
    world, rank, local_rank, dev, use_dist = setup(args)
    res = run_headline(args, world, rank, local_rank, dev, use_dist)
    out = None
    if rank == 0:
        out, (Y0, init0) = res
        # ---- CPU baseline on this host, bounded sample (rank 0, N = 1 only) ----
        if world == 1 and args.cpu_iters > 0:
            out['cpu_baseline'] = bb.cpu_baseline_em(Y0, init0, args.cpu_iters)
    # the blocks below are additions to the contract line: one that fails (a missing RCCL symbol, an
    # out-of-memory on a shared box) is recorded as {"error": ...} and must not cost the headline
    def guarded(name, fn):
        try:
            return fn()
        except Exception as e:   # noqa: BLE001 -- anything; the text goes into the line
            import traceback
            traceback.print_exc(file=sys.stderr)
            print(f'bench.py: block {name!r} failed, the line carries the error', file=sys.stderr)
            return {'error': f'{type(e).__name__}: {e}'[:300]}

    if rank == 0 and world == 1 and args.extras == 'auto':
        for name, fn in (('host_numpy_in_out', lambda: bb.pcie_inclusive(Y0, init0, args.iters)),
                         ('single_utterance_chain',
                          lambda: bb.single_utterance_chain(Y0, init0, args.iters, args.beamformer)),
                         ('canonical_call', lambda: bb.canonical_call(Y0))):
            out[name] = guarded(name, fn)
    if args.f32 == 'auto' and bb.has_f32():
        r32 = guarded('f32', lambda: run_headline(args, world, rank, local_rank, dev, use_dist,
                                                  precision='f32'))
        if rank == 0:
            if isinstance(r32, dict):
                out['f32'] = r32
            else:
                f32, _ = r32
                out['f32'] = {k: f32[k] for k in ('value', 'unit', 'ms_per_step', 'dtype',
                                                  'config', 'roofline', 'status_bits_or',
                                                  'sustained', 'verify') if k in f32}
    if args.config3 == 'auto':
        r3 = guarded('config3', lambda: bb.run_config3(args, world, rank, local_rank, dev,
                                                       use_dist))
        if rank == 0:
            out['config3'] = r3 if isinstance(r3, dict) else r3[0]
    if args.configs45 == 'auto' and world == 1:
        n45 = max(5, args.steps // 3)
        out['config4'] = guarded('config4', lambda: bb.run_config4(args, local_rank, dev, 'watson',
                                                                  steps=n45, warmup=2))
        out['config4']['vmf'] = guarded('config4.vmf', lambda: bb.run_config4(
            args, local_rank, dev, 'vmf', steps=n45, warmup=2))
        out['config5'] = guarded('config5', lambda: bb.run_config5(args, local_rank, dev,
                                                                  steps=n45, warmup=2))
    if rank == 0 and os.environ.get('PBBSS_BENCH_ONE_DEVICE') == '1':
        out['rehearsal'] = ('PBBSS_BENCH_ONE_DEVICE=1: all ranks shared GPU 0 over gloo -- a '
                            'functional rehearsal of the multi-rank path, not a measurement')
    emit(out if rank == 0 else None, use_dist, args)


if __name__ == '__main__':
    # run as a script: load this file ONCE more under its module name, so that tools/bench_blocks.py
    # (`import bench`) and this entry point share one copy of the helpers
    import bench
    bench.main()
