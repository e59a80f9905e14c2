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
  roofline     -- the resource that bounds the dominant kernel: float64 VALU
                  (useful flops of the kernel's algorithm over its duration,
                  measured with HIP events on the launch stream inside the
                  library, against the 78.6 TFLOP/s FP64 vector peak); the
                  SURVEY.md section 8d contract figure -- algorithmic HBM bytes
                  8*F*T*D per EM iteration against 8 TB/s -- rides along as
                  roofline.hbm_contract.  `traffic` (PMC FETCH_SIZE+WRITE_SIZE)
                  is only reported when a committed profile was taken from
                  exactly the kernel sources of this tree (source hash);
  cpu_baseline -- the reference itself (kind "reference") when /root/reference
                  is importable, else the NumPy oracle (kind "port"), timed on
                  this host on a bounded sample (rank 0, N = 1 only);
  mask_max_abs_err -- device vs oracle after all iterations on a bin subset.

`--workload config3` runs BASELINE configs[2] instead: a batch of 64 utterances
through EM -> mask all-gather -> DHTV alignment -> PSD -> gev+ban -> apply
(pb_bss_amd/pipeline.py), bins (`--shard bins`) or utterances (`--shard
utterances`) sharded over the ranks; one JSON line, strong scaling.
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
    p.add_argument('--workload', choices=['config2', 'config3'], default='config2',
                   help='config2: BASELINE configs[1] (headline); config3: configs[2], the batch '
                        'of utterances through EM + alignment + gev+ban')
    p.add_argument('--shard', choices=['bins', 'utterances'], default='bins',
                   help='config3 with N > 1: what is sharded over the ranks')
    p.add_argument('--utterances', type=int, default=64, help='config3 batch size')
    p.add_argument('--mask-gather', choices=['f64', 'f32'], default='f64',
                   help='config3 --shard bins: dtype of the mask all-gather')
    p.add_argument('--comm', choices=['torch', 'native'], default='torch',
                   help='mask all-gather through torch.distributed (RCCL) or through the '
                        'library\'s own RCCL communicator (C ABI pbbss_allgather_masks)')
    p.add_argument('--print-source-sha', action='store_true',
                   help='print the hash of the EM kernel sources (tools/profile_round.sh stamps '
                        'it into the profile summaries) and exit')
    return p.parse_args()


KERNEL_SOURCES = ('cacgmm_em.hpp', 'wave_la.hpp', 'pbbss_dev.hpp', 'em_inst.hip', 'em_launch.hpp')


def kernel_source_sha():
    """Hash of the sources the EM kernels are compiled from: a committed profile counts as
    evidence for the shipped kernel only if it carries the same hash."""
    import hashlib
    h = hashlib.sha1()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'pb_bss_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def matching_profile():
    """Newest profiles/r*_profile.txt taken from exactly these kernel sources, or None."""
    import glob
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_profile.txt')), reverse=True):
        with open(path) as f:
            head = f.read(2000)
        if f'kernel_source_sha: {sha}' in head:
            return path
    return None


def pmc_traffic():
    """HBM bytes per launch (main + concurrent split kernel) from the rocprofv3 PMC passes of
    this same command (tools/profile_round.sh -> profiles/rNN_x_profile.txt):
    (FETCH_SIZE + WRITE_SIZE) KiB.  PMC passes cannot run inside the timed bench, so the
    figure is read back from a committed summary -- and only from one taken from exactly the
    kernel sources of this tree; otherwise (None, reason)."""
    path = matching_profile()
    if path is None:
        return None, ('no committed profile carries kernel_source_sha ' + kernel_source_sha() +
                      ' (re-run tools/profile_round.sh on this tree)')
    kib = 0.0
    for line in open(path):
        parts = [x.strip() for x in line.split('|')]
        if len(parts) == 4 and parts[1] in ('FETCH_SIZE', 'WRITE_SIZE'):
            kib += float(parts[3])
    if kib == 0.0:
        return None, 'profiles/' + os.path.basename(path) + ' holds no FETCH_SIZE / WRITE_SIZE rows'
    return kib * 1024.0, ('profiles/' + os.path.basename(path) +
                          ': FETCH_SIZE + WRITE_SIZE of both kernels, separate --pmc passes; the '
                          'loads are 8 B/lane (not the 16 B/lane case the guide calibrates to x2): '
                          'calibrated on the kernel itself -- compulsory reads (Y + initialisation) '
                          '22.6 MB vs 21.2 MB counted')


def pmc_clock_ghz():
    """Shader clock during the profiled EM kernel: GRBM_GUI_ACTIVE (one counter per XCD, summed
    by rocprofv3) / 8 XCDs / kernel duration of that pass.  None without a matching profile."""
    path = matching_profile()
    if path is None:
        return None
    for line in open(path):
        parts = [x.strip() for x in line.split('|')]
        if len(parts) == 2 and parts[0] == 'main_clock_ghz':
            return float(parts[1])
    return None


def roofline_block(kernel_ms, bins, iters, world_note=''):
    """roofline object for the EM kernel: float64 VALU is what bounds it (the observation is
    LDS-resident, HBM traffic is 0.03x the contract figure), the section-8d HBM contract figure
    rides along."""
    avg_kernel_s = kernel_ms * 1e-3
    alg_bytes = 8.0 * bins * T * D * iters
    hbm = alg_bytes / avg_kernel_s / 1e9
    tflops = FLOPS_PER_FRAME_ITER * bins * T * iters / avg_kernel_s / 1e12
    traffic, traffic_src = pmc_traffic() if (bins == F and iters == 100) else (
        None, 'PMC traffic is only collected for the single-utterance headline command')
    clock = pmc_clock_ghz()
    return {
        'bound': 'fp64_valu', 'achieved': tflops, 'peak': FP64_VALU_PEAK_TF, 'unit': 'TFLOP/s',
        'frac': tflops / FP64_VALU_PEAK_TF, 'traffic': traffic, 'traffic_source': traffic_src,
        'kernel': 'cacgmm_em_kernel<8,3,float,false> (+ concurrent cacgmm_em_split_kernel for the '
                  'remainder bins of a launch; kernel_ms brackets both)' + world_note,
        'kernel_ms': kernel_ms,
        'flops_per_frame_iter': FLOPS_PER_FRAME_ITER,
        'flops_note': 'useful float64 flops of the kernel\'s algorithm per frame and EM iteration: '
                      'Hermitian outer product P in the E phase (192) and again in the M phase '
                      '(192, it cannot stay resident: 256 KB per bin), q_k = <A_k, P> (2 D^2 K), '
                      'C_k += w_k P (2 D^2 K), softmax (~100)',
        'peak_note': 'MI355X FP64 vector peak at 2.4 GHz; under this load the chip sustains '
                     'clock_ghz (power management), see frac_at_sustained_clock',
        'clock_ghz': clock,
        'frac_at_sustained_clock': None if not clock else tflops / (FP64_VALU_PEAK_TF * clock / 2.4),
        'hbm_contract': {
            'bound': 'hbm', 'achieved': hbm, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': hbm / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': alg_bytes,
            'note': 'SURVEY.md section 8d: 8*F*T*D bytes per EM iteration (one read of the '
                    'complex64 observation) against 8 TB/s; the kernel keeps y in LDS for the '
                    'whole EM loop, so HBM does not bind',
        },
    }


def cpu_baseline_em(Y0, init0, iters):
    """EM iterations/s of the CPU path on this host: the unmodified reference when it is
    importable here (/root/reference through the tests' shim), else the NumPy oracle."""
    ref_dir = '/root/reference'
    if os.path.isdir(os.path.join(ref_dir, 'pb_bss')):
        try:
            from oracle import refshim
            refshim.load()
            from pb_bss.distribution import CACGMMTrainer as RefTrainer
            runs = []
            for dtype, label in ((np.complex128, 'float64 path (complex128 input)'),
                                 (np.complex64, 'float32 path (complex64 input + ndarray init)')):
                best = None
                for _ in range(3):
                    t1 = time.perf_counter()
                    RefTrainer().fit(Y0.astype(dtype), initialization=init0, iterations=max(iters // 10, 2))
                    dt = time.perf_counter() - t1
                    best = dt if best is None else min(best, dt)
                runs.append((label, max(iters // 10, 2) / best))
            return {
                'value': runs[0][1], 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'reference',
                'sample': f'pb_bss CACGMMTrainer.fit imported from {ref_dir}, full F=513 T=500 D=8 '
                          f'K=3, {max(iters // 10, 2)} EM iterations, best of 3: ' +
                          '; '.join(f'{l}: {v:.2f} it/s' for l, v in runs) +
                          f'; host has {os.cpu_count()} logical cores, einsum is single-threaded',
            }
        except Exception as e:  # fall through to the oracle, say why
            note = f' (reference import failed: {type(e).__name__}: {e})'
    else:
        note = ' (no /root/reference on this host)'
    from oracle import cacgmm as oc
    Y128 = Y0.astype(np.complex128)
    t1 = time.perf_counter()
    oc.em_fit(Y128, init0, iterations=iters)
    dt = time.perf_counter() - t1
    return {
        'value': iters / dt, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
        'sample': f'NumPy oracle (oracle/cacgmm.py: restated einsum contractions, float64; it forms '
                  f'B^-1 first and measures ~1.5x faster than the reference\'s own calls), full '
                  f'F=513 T=500 D=8 K=3, {iters} EM iterations, {dt:.1f} s; host has '
                  f'{os.cpu_count()} logical cores, einsum is single-threaded' + note,
    }


def emit(line, use_dist):
    import torch.distributed as dist
    if use_dist:
        from pb_bss_amd import sharding
        sharding.destroy_native_comm()
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


def setup(args):
    import torch
    import torch.distributed as dist
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
        if args.comm == 'native':
            from pb_bss_amd import sharding
            sharding.init_native_comm(device_index=local_rank)
    return world, rank, local_rank, dev, use_dist


def timed(step, args, use_dist, dev):
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize; max over ranks."""
    import torch
    import torch.distributed as dist

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    last = None
    kernel_ms = 0.0
    for _ in range(args.steps):
        last = step()
        kernel_ms += last['kernel_ms']
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, kernel_ms / args.steps, last


def main_config3(args):
    """BASELINE configs[2]: `--utterances` utterances of config 2 through the whole chain."""
    import torch
    from pb_bss_amd.testing import synth
    from pb_bss_amd import _lib, engine, pipeline
    world, rank, local_rank, dev, use_dist = setup(args)
    U = args.utterances
    data = [synth.make_stft(F, T, D, K, seed=u) for u in range(U)]
    Y = _lib.to_device(np.stack([d[0] for d in data]))          # (U, F, T, D) complex64
    init = _lib.to_device(np.stack([d[1] for d in data]))       # (U, F, K, T) float64
    shard = args.shard if use_dist and world > 1 else None
    gdt = torch.float32 if args.mask_gather == 'f32' else None
    engine.set_timing(True, local_rank)

    def step():
        out = pipeline.separate(Y, init, args.iters, 2 * (F - 1), shard=shard,
                                mask_gather_dtype=gdt)
        out['kernel_ms'] = 0.0  # per-stage times are taken in a separate, untimed pass below
        return out

    elapsed, _, out = timed(step, args, use_dist, dev)
    line = None
    if rank == 0:
        # untimed pass with a synchronisation after every stage: where the step time goes
        stages = {}
        if shard is None:
            from pb_bss_amd.pipeline import device_ops as ops, _chain_after_masks

            def lap(name, fn):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                stages[name] = (time.perf_counter() - t1) * 1e3
                return r
            masks = lap('em_fit_predict_ms', lambda: ops.em_masks(Y, init, args.iters))
            em_kernel_ms = engine.last_kernel_ms(local_rank)
            mapping = lap('dhtv_mapping_ms', lambda: ops.dhtv_mapping(
                masks.transpose(-3, -2).contiguous(), 2 * (F - 1)))
            lap('align_psd_gev_ban_apply_ms', lambda: _chain_after_masks(Y, masks, mapping, ops))
        else:
            em_kernel_ms = None
        res = {
            'metric': 'cACGMM EM iterations/sec on F=513,T=500,D=8,K=3',
            'value': U * args.iters * args.steps / elapsed,
            'unit': 'EM iterations/s (utterance-iterations, whole job; every step also runs DHTV '
                    'alignment, PSD, gev+ban and apply for all utterances)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': f'BASELINE configs[2]: batch of {U} utterances, 8-mic 3-source cACGMM '
                            f'(F=513 T=500 D=8 K=3, {args.iters} EM iterations) + DHTV permutation '
                            f'alignment + PSD + gev+ban beamformer + apply, complex64 STFTs resident '
                            f'in HBM',
                'utterances': U, 'em_iterations_per_step': args.iters,
                'sharding': ('none (1 GPU)' if shard is None else
                             f'{shard} over {world} ranks' +
                             (f'; one RCCL all-gather of the masks ({args.mask_gather}) + one of the '
                              f'(U, K, F) mappings per step' if shard == 'bins' else
                              '; no collective')),
                'utterances_per_s': U * args.steps / elapsed,
            },
            'stage_ms_untimed_pass': stages or None,
        }
        if em_kernel_ms:
            res['roofline'] = roofline_block(em_kernel_ms, U * F, args.iters,
                                             f'; here one launch over {U * F} bins, three workgroups per CU')
        if args.check_bins:
            from oracle import cacgmm as oc, permutation_alignment as op
            nb = min(args.check_bins, F)
            Y0 = data[0][0][:nb].astype(np.complex128)
            ref = oc.em_predict(oc.em_fit(Y0, data[0][1][:nb], iterations=args.iters), Y0)  # (nb,K,T)
            mapping0 = _lib.to_host(out['mapping'])[0]                                      # (K, F)
            got = _lib.to_host(out['masks'])[0]                                             # (K, F', T)
            # undo the alignment on the checked bins: aligned[k, f] = masks[mapping[k, f], f]
            err = 0.0
            for f in range(min(nb, got.shape[1])):
                err = max(err, float(np.abs(got[:, f] - ref[f][mapping0[:, f]]).max()))
            res['mask_max_abs_err'] = err
            res['mask_err_bins_checked'] = nb
        if world == 1 and args.cpu_iters > 0:
            from oracle import beamformer as ob, cacgmm as oc, permutation_alignment as op
            Y128 = data[0][0].astype(np.complex128)
            t1 = time.perf_counter()
            m = oc.em_predict(oc.em_fit(Y128, data[0][1], iterations=args.cpu_iters), Y128)
            kft = m.transpose(1, 0, 2)
            plan = op.alignment_plan(2 * (F - 1), **op.PRESETS[2 * (F - 1)])
            al = op.apply_mapping(kft, op.dhtv_calculate_mapping(kft, plan))
            X = Y128.transpose(0, 2, 1)
            psd = ob.psd(X, al.transpose(1, 0, 2))
            for k in range(K):
                ob.apply_bf(ob.bf_vector('gev+ban', psd[:, k], psd.sum(1) - psd[:, k]), X)
            dt = time.perf_counter() - t1
            res['cpu_baseline'] = {
                'value': args.cpu_iters / dt, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
                'sample': f'NumPy oracle chain (EM {args.cpu_iters} iterations + DHTV + gev+ban + '
                          f'apply) on ONE of the {U} utterances, {dt:.1f} s; host has '
                          f'{os.cpu_count()} logical cores',
            }
        line = json.dumps(res)
    emit(line, use_dist)


def main():
    args = parse()
    if args.print_source_sha:
        print(kernel_source_sha())
        return
    if args.workload == 'config3':
        return main_config3(args)
    import torch
    import torch.distributed as dist
    from pb_bss_amd.testing import synth  # input generator shared with the parity tests
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.sharding import all_gather_bins, shard_bounds
    world, rank, local_rank, dev, use_dist = setup(args)

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
        return {'masks': masks, 'kernel_ms': ms, 'r': r}

    elapsed, kernel_ms, last = timed(step, args, use_dist, dev)
    masks, r = last['masks'], last['r']

    line = None
    if rank == 0:
        out = {
            'metric': 'cACGMM EM iterations/sec on F=513,T=500,D=8,K=3',
            'value': world * args.iters * args.steps / elapsed,
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
                             'all-gather per step (' + args.comm + ' communicator), in stream order after the EM kernel') if use_dist else 'none (1 GPU)',
            },
            'roofline': roofline_block(kernel_ms, world * n_loc, args.iters),
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
        # ---- CPU baseline on this host, bounded sample ------------------------
        if world == 1 and args.cpu_iters > 0:
            out['cpu_baseline'] = cpu_baseline_em(Y0, init0, args.cpu_iters)
        line = json.dumps(out)
    emit(line, use_dist)


if __name__ == '__main__':
    main()
