"""Workload blocks of bench.py beyond the headline: BASELINE configs[2] (64 utterances through the
chain), configs[3] (Watson / vMF mixtures + MVDR-Souden), configs[4] (joint GCACGMM), the secondary
single-utterance figures, the CPU baselines and the reader of the committed rocprofv3 summaries.
bench.py holds the timing contract (fence / timed / max over ranks), the headline workload and the
compact emitter; everything here goes to `bench_extra.json` in full and to the driver's line as
one-number summaries (bench.compact_line)."""
import json
import math
import os
import sys
import time

import numpy as np

import bench
from bench import (ROOT, F, T, D, K, HBM_PEAK_GBS, FP64_VALU_PEAK_TF, FP32_VALU_PEAK_TF,
                   FLOPS_PER_FRAME_ITER, emit, setup, fence, max_over_ranks, preheat, run_steps, timed,
                   sustained, roofline_block)

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


def read_profile(path):
    """Rows of a tools/profile_round.sh summary: {'pmc': {(kernel, counter): mean}, 'trace':
    {'median_us', 'min_us', 'n'} of the measured EM-kernel launches, 'clock_ghz'}."""
    out = {'pmc': {}, 'trace': None, 'clock_ghz': None}
    for line in open(path):
        parts = [x.strip() for x in line.split('|')]
        if len(parts) == 4 and parts[1].isupper() and parts[2].isdigit():
            try:
                out['pmc'][(parts[0], parts[1])] = float(parts[3])
            except ValueError:
                pass
        elif len(parts) == 5 and parts[0] == 'em_kernel_trace_us':
            out['trace'] = {'median_us': float(parts[1]), 'min_us': float(parts[2]),
                            'max_us': float(parts[3]), 'n': int(parts[4])}
        elif len(parts) == 2 and parts[0] == 'main_clock_ghz':
            out['clock_ghz'] = float(parts[1])
    return out


def pmc_evidence(ms_per_step):
    """-> (traffic_bytes or None, clock_ghz or None, source / reason).  PMC passes cannot run
    inside the timed bench, so the figures are read back from a committed summary -- only from one
    taken from exactly the kernel sources of this tree, and only if that profile's own kernel
    trace is consistent with this run: its median EM-kernel duration must not exceed this run's
    ms_per_step by more than 2 % (a trace slower than the un-profiled step is not evidence)."""
    path = matching_profile()
    if path is None:
        return None, None, ('no committed profile carries kernel_source_sha ' + kernel_source_sha() +
                            ' (re-run tools/profile_round.sh on this tree)')
    name = 'profiles/' + os.path.basename(path)
    prof = read_profile(path)
    if prof['trace'] is None:
        return None, None, name + ' holds no em_kernel_trace_us row (old format)'
    med_ms = prof['trace']['median_us'] * 1e-3
    if ms_per_step is not None and med_ms > 1.02 * ms_per_step:
        return None, None, (f'{name}: kernel-trace median {med_ms:.4f} ms exceeds this run\'s '
                            f'ms_per_step {ms_per_step:.4f} by more than 2 %: not used as evidence')
    kib = sum(v for (k, c), v in prof['pmc'].items() if c in ('FETCH_SIZE', 'WRITE_SIZE'))
    if kib == 0.0:
        return None, prof['clock_ghz'], name + ' holds no FETCH_SIZE / WRITE_SIZE rows'
    # gfx950: FETCH_SIZE reports half the bytes of coalesced streaming reads (MI355X_MICROARCH.md;
    # calibrated here on embed_prepare_kernel's known 41.04 MB read: 20.06 MB counted,
    # profiles/r04_l_config5_profile.txt) -- doubled; WRITE_SIZE is taken as counted (the same
    # kernel's 41.04 MB written: 43.6 MB counted)
    fetch_raw = sum(v for (k, c), v in prof['pmc'].items() if c == 'FETCH_SIZE') * 1024.0
    write = sum(v for (k, c), v in prof['pmc'].items() if c == 'WRITE_SIZE') * 1024.0
    compulsory = 8.0 * F * T * D + 8.0 * F * K * T  # one read of Y (complex64) + the initialisation
    return 2.0 * fetch_raw + write, prof['clock_ghz'], (
        f'{name}: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE of the EM and the split kernel, '
        f'separate --pmc passes, per launch; kernel-trace median {med_ms:.4f} ms over '
        f'{prof["trace"]["n"]} measured launches; reads {2.0 * fetch_raw / 1e6:.1f} MB (counted '
        f'{fetch_raw / 1e6:.1f}) vs {compulsory / 1e6:.1f} MB compulsory (Y + initialisation), '
        f'writes {write / 1e6:.1f} MB vs {8.0 * F * K * T / 1e6:.1f} MB of masks')


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
            n = max(iters // 10, 2)
            for dtype, label in ((np.complex128, 'float64 path (complex128 input)'),
                                 (np.complex64, 'float32 path (complex64 input + ndarray init)')):
                Yd = Y0.astype(dtype)
                med, _ = median_rate(lambda: RefTrainer().fit(Yd, initialization=init0,
                                                              iterations=n), n)
                runs.append((label, med))
            return {
                'value': runs[0][1], 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'reference',
                'sample': f'pb_bss CACGMMTrainer.fit imported from {ref_dir}, full F=513 T=500 D=8 '
                          f'K=3, {n} EM iterations, median of 3 runs: ' +
                          '; '.join(f'{l}: {v:.2f} it/s' for l, v in runs) +
                          f'; host has {os.cpu_count()} logical cores, einsum is single-threaded',
                'reference_recorded': reference_recorded('config2'),
            }
        except Exception as e:  # fall through to the oracle, say why
            note = f' (reference import failed: {type(e).__name__}: {e})'
    else:
        note = ' (no /root/reference on this host)'
    from oracle import cacgmm as oc
    Y128 = Y0.astype(np.complex128)
    n = max(iters // 3, 2)
    t1 = time.perf_counter()
    med, runs = median_rate(lambda: oc.em_fit(Y128, init0, iterations=n), n)
    dt = time.perf_counter() - t1
    return {
        'value': med, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port', 'runs': runs,
        'timing_mode': PORT_NOTE,
        'sample': f'NumPy oracle (oracle/cacgmm.py, float64, reference-shaped timing mode), full '
                  f'F=513 T=500 D=8 K=3, median of 3 runs of {n} EM iterations, {dt:.1f} s in all; '
                  f'host has {os.cpu_count()} logical cores, 1 used (einsum is single-threaded; the '
                  f'reference would use 1 core here too)' + note,
        'reference_recorded': reference_recorded('config2'),
    }


def pcie_inclusive(Y0, init0, iters, reps=10):
    """The drop-in call a pb_bss user makes: NumPy arrays in, NumPy masks out, through
    CACGMMTrainer.fit_predict (H2D of the observation and the initialisation, the fit, D2H of the
    masks).  A secondary figure: `value` is measured with the inputs resident in HBM."""
    from pb_bss_amd.distribution import CACGMMTrainer
    tr = CACGMMTrainer()
    tr.fit_predict(Y0, initialization=init0, iterations=iters)  # first call: allocations
    t0 = time.perf_counter()
    for _ in range(reps):
        masks = tr.fit_predict(Y0, initialization=init0, iterations=iters)
    dt = (time.perf_counter() - t0) / reps
    assert isinstance(masks, np.ndarray)
    nbytes = Y0.nbytes + init0.nbytes + masks.nbytes
    return {'ms_per_call': dt * 1e3, 'value': iters / dt, 'unit': 'EM iterations/s',
            'bytes_over_pcie_per_call': int(nbytes),
            'what': 'CACGMMTrainer().fit_predict(Y, initialization=init) with NumPy arrays in and '
                    'out (complex64 observation + float64 initialisation up, float64 masks down); '
                    'never reported as `value`'}


def canonical_call(Y0, reps=10):
    """The reference's canonical call (examples/mixture_model_example.ipynb, cacgmm.py:205-210):
    `CACGMMTrainer().fit(Y, num_classes=3)` with a RANDOM initialisation, 10 EM iterations, NumPy
    observation in, model out.  'numpy' draws the initialisation from NumPy's global generator as
    the reference does (host RNG + upload on the critical path); 'device' is the opt-in GPU draw
    (pb_bss_amd.distribution.utils.set_random_init).  Secondary figures."""
    import torch
    from pb_bss_amd.distribution import CACGMMTrainer, utils
    from pb_bss_amd import _lib
    out = {}
    Yd = _lib.to_device(Y0)
    for mode in ('numpy', 'device'):
        with utils.random_init(mode):
            for resident, Yin in (('numpy_in', Y0), ('resident', Yd)):
                CACGMMTrainer().fit(Yin, num_classes=K, iterations=10)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    CACGMMTrainer().fit(Yin, num_classes=K, iterations=10)
                torch.cuda.synchronize()
                out[f'{mode}_init_{resident}_ms'] = (time.perf_counter() - t0) / reps * 1e3
    # cacgmm.py:260-267: the DHTV aligner between E- and M-step of every iteration (step-wise loop)
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    try:
        aligner = DHTVPermutationAlignment.from_stft_size(2 * (Y0.shape[0] - 1))
    except ValueError:  # no preset for this STFT size
        aligner = None
    if aligner is not None:
        rng = np.random.default_rng(11)
        g0 = rng.uniform(size=(Y0.shape[0], K, Y0.shape[1]))
        g0 = _lib.to_device(g0 / g0.sum(1, keepdims=True))
        kw = dict(weight_constant_axis=(-3,), inline_permutation_aligner=aligner)
        settled = CACGMMTrainer().fit(Yd, initialization=g0, iterations=40, **kw)
        for tag, start, n in (('inline_aligner_first_20_ms_per_iteration', g0, 20),
                              ('inline_aligner_settled_ms_per_iteration', settled, 20)):
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                CACGMMTrainer().fit(Yd, initialization=start, iterations=n, **kw)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                best = dt if best is None else min(best, dt)
            out[tag] = best
    out['what'] = ('CACGMMTrainer().fit(Y, num_classes=3, iterations=10), ms per call: random '
                   'initialisation drawn by NumPy\'s global generator (the reference\'s stream; '
                   'default) or on the device (opt-in, a different stream), observation given as a '
                   'NumPy array or already resident; inline_aligner_*: fit(..., weight_constant_axis='
                   '(-3,), inline_permutation_aligner=DHTVPermutationAlignment.from_stft_size(...)), '
                   'ms per EM iteration from a random start and resumed after 40 iterations')
    return out


def single_utterance_chain(Y0, init0, iters, beamformer, reps=10):
    """Latency of the chain a caller runs on ONE utterance with everything resident in HBM: EM fit +
    predict, DHTV permutation alignment of the masks, PSD -> beamformer -> apply.  Wall clock per
    stage with a synchronisation in between (so the stages add up to slightly more than a chained
    call); a secondary figure next to `value`."""
    import torch
    from pb_bss_amd import _lib
    from pb_bss_amd.pipeline import device_ops as ops, _chain_after_masks
    Y = _lib.to_device(Y0)[None]
    init = _lib.to_device(init0)[None]
    Fb = Y.shape[-3]
    stages = {'em_fit_predict_ms': [], 'dhtv_mapping_ms': [], 'align_psd_bf_apply_ms': []}

    def lap(name, fn):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        stages[name].append((time.perf_counter() - t1) * 1e3)
        return r
    for _ in range(reps + 2):
        masks = lap('em_fit_predict_ms', lambda: ops.em_masks(Y, init, iters))
        mapping = lap('dhtv_mapping_ms', lambda: ops.dhtv_mapping(
            masks.transpose(-3, -2).contiguous(), 2 * (Fb - 1)))
        lap('align_psd_bf_apply_ms', lambda: _chain_after_masks(Y, masks, mapping, ops, beamformer))
    med = {k: float(np.median(v[2:])) for k, v in stages.items()}
    med['total_ms'] = float(sum(med.values()))
    med['what'] = ('one utterance, inputs resident in HBM, median of %d runs per stage with a '
                   'synchronisation after each: EM (%d iterations) + predict, DHTV alignment of the '
                   'masks, alignment + PSD + %s + apply' % (reps, iters, beamformer))
    return med


def checksum(x):
    """Three float64 numbers that pin a tensor bit for bit in practice (sum, sum of squares,
    position-weighted sum) -- compared across ranks to prove they hold identical gathered data."""
    import torch
    v = x.reshape(-1).to(torch.float64)
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.float64) % 1021.0) + 1.0
    return torch.stack([v.sum(), (v * v).sum(), (v * w).sum()])


def identical_on_all_ranks(x, use_dist, world):
    import torch
    import torch.distributed as dist
    c = checksum(x)
    if not use_dist:
        return True
    allc = torch.empty((world, 3), dtype=torch.float64, device=c.device)
    dist.all_gather_into_tensor(allc, c.unsqueeze(0).contiguous())
    return bool((allc == allc[0:1]).all().item())


def per_rank_table(values, use_dist, world, dev):
    """{name: float} of THIS rank -> {name: [value of rank 0, rank 1, ...]} on every rank (one
    small all-gather; at N = 1 a one-element list each).  What a first scaling run needs in order
    to explain itself: which rank, and which stage of it, sets the max-over-ranks time."""
    import torch
    import torch.distributed as dist
    names = sorted(values)
    mine = torch.tensor([float(values[n]) for n in names], dtype=torch.float64, device=dev)
    if use_dist:
        allv = torch.empty((world, len(names)), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allv, mine.unsqueeze(0).contiguous())
    else:
        allv = mine.unsqueeze(0)
    allv = allv.cpu().tolist()
    return {n: [round(allv[r][i], 4) for r in range(len(allv))] for i, n in enumerate(names)}


def spread(lo, hi, n):
    """n bins of [lo, hi) including the first and the last."""
    if hi - lo <= n:
        return list(range(lo, hi))
    return sorted({lo + int(round(i * (hi - 1 - lo) / (n - 1))) for i in range(n)})


def comm_block(args, use_dist, world, bytes_per_rank_step, gather_ms, what):
    if not use_dist:
        return None
    import ctypes
    import torch.distributed as dist
    from pb_bss_amd import _lib, sharding
    blk = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
           'communicator': args.comm, 'what': what,
           'bytes_received_per_rank_per_step': bytes_per_rank_step, 'gather_ms': gather_ms}
    if gather_ms:
        blk['gather_GBps_per_rank'] = bytes_per_rank_step / (gather_ms * 1e-3) / 1e9
    if args.comm == 'native' and sharding.native_comm() is not None:
        w, r = ctypes.c_int(-1), ctypes.c_int(-1)
        if _lib.load().pbbss_comm_info(_lib.handle(), ctypes.byref(w), ctypes.byref(r)) == 0:
            blk['rccl_comm_count'] = w.value  # ncclCommCount of the library's communicator
    return blk


# ------------------------------------------------------------------------------------------
# BASELINE configs[2]: a batch of utterances through the whole chain
# ------------------------------------------------------------------------------------------
def make_batch(U):
    """U seeded utterances of config 2 (seed = utterance index), generated on a few threads."""
    from concurrent.futures import ThreadPoolExecutor
    from pb_bss_amd.testing import synth
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda u: synth.make_stft(F, T, D, K, seed=u), range(U)))


_ORACLE_CACHE = {}


def oracle_bins(data, u, fs, iters):
    """Oracle masks (nb, K, T) of bins `fs` of utterance u after `iters` iterations (cached:
    the bins- and the utterances-sharded legs check the same bins)."""
    from oracle import cacgmm as oc
    key = (u, tuple(fs), iters)
    if key not in _ORACLE_CACHE:
        Y128 = data[u][0][fs].astype(np.complex128)
        _ORACLE_CACHE[key] = oc.em_predict(oc.em_fit(Y128, data[u][1][fs], iterations=iters), Y128)
    return _ORACLE_CACHE[key]


def config3_leg(args, data, Y, init, shard, world, rank, dev, use_dist, steps, warmup):
    """One measurement of pipeline.separate on the whole batch; returns the result dict on rank 0."""
    import torch
    from pb_bss_amd import _lib, pipeline
    from pb_bss_amd.sharding import shard_bounds
    U = Y.shape[0]
    gdt = torch.float32 if args.mask_gather == 'f32' else None

    def step():
        return pipeline.separate(Y, init, args.iters, 2 * (F - 1), shard=shard,
                                 mask_gather_dtype=gdt, beamformer=args.beamformer)

    elapsed, _, _ = timed(step, steps, warmup, use_dist, dev)
    # ---- untimed verification pass: outputs of ALL ranks gathered ----
    out = pipeline.separate(Y, init, args.iters, 2 * (F - 1), shard=shard, mask_gather_dtype=gdt,
                            beamformer=args.beamformer, gather_output=shard is not None)
    # ---- untimed diagnostic pass: this rank's wall time per stage, synchronised after each ----
    stage = {}
    pipeline.separate(Y, init, args.iters, 2 * (F - 1), shard=shard, mask_gather_dtype=gdt,
                      beamformer=args.beamformer, stage_ms=stage)
    for name in ('em_ms', 'gather_ms', 'dhtv_ms', 'map_gather_ms', 'extract_ms'):
        stage.setdefault(name, 0.0)
    stage_tab = per_rank_table(stage, use_dist and shard is not None, world, dev)
    map_same = identical_on_all_ranks(out['mapping'], use_dist and shard is not None, world)
    enh_same = identical_on_all_ranks(torch.view_as_real(out['enhanced'].contiguous()),
                                      use_dist and shard is not None, world)
    if rank != 0:
        return None
    res = {
        'value': U * args.iters * steps / elapsed,
        'unit': 'EM iterations/s (utterance-iterations, whole job; every step also runs DHTV '
                'alignment, PSD, ' + args.beamformer + ' and apply for all utterances)',
        'ms_per_step': elapsed / steps * 1e3, 'steps': steps, 'warmup': warmup,
        'utterances_per_s': U * steps / elapsed, 'scaling': 'strong', 'n_gpus': world,
        'per_rank_stage_ms': dict(stage_tab, what='untimed pass with a device synchronisation '
                                                  'after every stage (the stages of the timed '
                                                  'steps overlap, these do not): EM + predict, '
                                                  'mask all-gather, DHTV mapping of the rank\'s '
                                                  'utterances, mapping all-gather, alignment + '
                                                  'PSD + beamformer + apply'),
        'sharding': ('none (1 GPU)' if shard is None else
                     f'{shard} over {world} ranks' +
                     (f'; one RCCL all-gather of the masks ({args.mask_gather}) + one of the '
                      f'(U, K, F) mappings per step' +
                      ('; one all-reduce of the Souden reference-channel sums'
                       if args.beamformer == 'mvdr_souden' else '')
                      if shard == 'bins' else '; no collective')),
    }
    if shard == 'bins' and use_dist:
        per_rank = U * F * K * T * (4 if gdt is not None else 8) * (world - 1) // world
        res['comm'] = comm_block(args, use_dist, world, per_rank, None,
                                 'mask all-gather + (U, K, F) int64 mapping all-gather per step')
    if args.check_bins:
        from oracle import beamformer as ob
        # one utterance of every rank's utterance share, bins of every rank's bin shard
        us = sorted({shard_bounds(U, world, rr)[0] for rr in range(world)} | {U - 1})
        if world == 1:
            us = sorted({0, U - 1})
        nb = max(2, 12 // world)
        fs = sorted({f for rr in range(world) for f in spread(*shard_bounds(F, world, rr), nb)})
        mapping = _lib.to_host(out['mapping'])                     # (U, K, F)
        got = _lib.to_host(out['masks'])                           # (U, K, F, T) aligned
        enh = _lib.to_host(out['enhanced'])                        # (U, K, F, T) complex
        m_err = e_err = 0.0
        perm_ok = True
        for u in us:
            ref = oracle_bins(data, u, fs, args.iters)             # (nb, K, T)
            X = data[u][0][fs].astype(np.complex128).transpose(0, 2, 1)
            al = np.stack([ref[j][mapping[u][:, f]] for j, f in enumerate(fs)])   # (nb, K, T)
            for j, f in enumerate(fs):
                perm_ok = perm_ok and sorted(mapping[u][:, f].tolist()) == list(range(K))
                m_err = max(m_err, float(np.abs(got[u][:, f] - al[j]).max()))
            if args.beamformer == 'gev+ban':
                psd = ob.psd(X, al)                                # (nb, K, D, D)
                for k in range(K):
                    w = ob.bf_vector('gev+ban', psd[:, k], psd.sum(1) - psd[:, k])
                    s = np.abs(ob.apply_bf(w, X))                  # GEV phase is arbitrary: moduli
                    d = np.abs(np.abs(enh[u][k][fs]) - s).max() / max(float(s.max()), 1e-300)
                    e_err = max(e_err, float(d))
        res['verify'] = {
            'utterances_checked': us, 'bins_checked_per_utterance': fs,
            'mask_max_abs_err': m_err, 'enhanced_modulus_max_rel_err':
                e_err if args.beamformer == 'gev+ban' else None,
            'mapping_columns_are_permutations': perm_ok,
            'mapping_identical_on_all_ranks': map_same,
            'enhanced_identical_on_all_ranks': enh_same,
            'tolerance': 1e-5, 'ok': bool(m_err < 1e-5 and e_err < 1e-5 and perm_ok and map_same),
            'what': 'aligned masks of bins from every rank\'s bin shard, of one utterance of every '
                    'rank\'s utterance share, vs the NumPy oracle EM with the device mapping '
                    'applied; |enhanced| vs the oracle PSD -> gev+ban -> apply on those bins',
        }
        res['_mapping_checksum'] = [float(v) for v in checksum(out['mapping']).tolist()]
    return res


def run_config3(args, world, rank, local_rank, dev, use_dist, primary=False):
    """All config-3 legs of this run -> dict (rank 0) or None."""
    from pb_bss_amd import _lib
    import torch
    U = args.utterances
    data = make_batch(U)
    # the legs before this one leave torch's caching allocator full of blocks of other sizes; the
    # 0.4 GB temporaries of a 64-utterance step would then be carved by fresh hipMallocs inside the
    # timed steps (measured: 87 instead of 82 ms per step) -- start from an empty cache, warm up twice
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    Y = _lib.to_device(np.stack([d[0] for d in data]))          # (U, F, T, D) complex64
    init = _lib.to_device(np.stack([d[1] for d in data]))       # (U, F, K, T) float64
    steps = args.steps if primary else args.config3_steps
    warmup = args.warmup if primary else 2
    legs = {}
    if use_dist and world > 1:
        shards = [args.shard] if primary else ['bins', 'utterances']
        for sh in shards:
            # (keys 'bins_sharded' / 'utterances_sharded': 'utterances' is the batch size of the block)
            legs[sh + '_sharded'] = config3_leg(args, data, Y, init, sh, world, rank, dev, use_dist,
                                                steps, warmup)
    else:
        legs['single'] = config3_leg(args, data, Y, init, None, world, rank, dev, use_dist, steps,
                                     warmup)
        if world == 1 and rank == 0 and legs['single'] is not None:
            from pb_bss_amd import engine
            engine.set_timing(True, local_rank)
            config3_extras(args, data, Y, init, legs['single'], local_rank)
    if rank != 0:
        return None, data
    blk = {
        'workload': f'BASELINE configs[2]: batch of {U} utterances, 8-mic 3-source cACGMM '
                    f'(F=513 T=500 D=8 K=3, {args.iters} EM iterations) + DHTV permutation '
                    f'alignment + PSD + {args.beamformer} beamformer + apply, complex64 STFTs '
                    f'resident in HBM',
        'utterances': U, 'em_iterations_per_step': args.iters, 'scaling': 'strong',
    }
    sums = {k: v.pop('_mapping_checksum', None) for k, v in legs.items() if v}
    if len(sums) == 2 and all(sums.values()):
        a, b = list(sums.values())
        blk['mapping_identical_across_shardings'] = bool(a == b)
    blk.update(legs)
    return blk, data


def config3_stage_times(args, Y, init, local_rank):
    """Untimed pass with a synchronisation after every stage: where the step time goes (1 GPU)."""
    import torch
    from pb_bss_amd import engine
    from pb_bss_amd.pipeline import device_ops as ops, _chain_after_masks
    stages = {}

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
    lap('align_psd_bf_apply_ms', lambda: _chain_after_masks(Y, masks, mapping, ops, args.beamformer))
    return stages, em_kernel_ms


def config3_cpu_baseline(args, data, U):
    """The NumPy oracle chain (EM + DHTV + PSD + gev+ban + apply) on ONE utterance, median of 3."""
    from oracle import beamformer as ob, cacgmm as oc, permutation_alignment as op
    n = max(args.cpu_iters // 3, 2)
    Y128 = data[0][0].astype(np.complex128)

    def chain():
        m = oc.em_predict(oc.em_fit(Y128, data[0][1], iterations=n), Y128)
        kft = m.transpose(1, 0, 2)
        plan = op.alignment_plan(2 * (F - 1), **op.PRESETS[2 * (F - 1)])
        al = op.apply_mapping(kft, op.dhtv_calculate_mapping(kft, plan))
        X = Y128.transpose(0, 2, 1)
        psd = ob.psd(X, al.transpose(1, 0, 2))
        for k in range(K):
            ob.apply_bf(ob.bf_vector('gev+ban', psd[:, k], psd.sum(1) - psd[:, k]), X)

    t1 = time.perf_counter()
    med, runs = median_rate(chain, n)
    dt = time.perf_counter() - t1
    return {
        'value': med, 'unit': 'EM iterations/s (utterance-iterations; every run also pays DHTV '
                              'alignment, PSD, gev+ban and apply once)',
        'cores': 1, 'kind': 'port', 'runs': runs, 'timing_mode': PORT_NOTE,
        'sample': f'NumPy oracle chain (oracle/: EM {n} iterations + final E-step + DHTV alignment + '
                  f'PSD + gev+ban + apply) on ONE of the {U} utterances, median of 3 runs, {dt:.1f} s '
                  f'in all; host has {os.cpu_count()} logical cores, 1 used',
        'reference_recorded': reference_recorded('config2'),
    }


def config3_extras(args, data, Y, init, leg, local_rank):
    """roofline (EM stage: FP64 VALU + the section-8d HBM contract figure; the same two fractions
    over the WHOLE step; PMC traffic of the whole step from a committed profile of exactly these
    sources) and cpu_baseline for the one-GPU config-3 measurement `leg` (in place)."""
    U = args.utterances
    stages, em_kernel_ms = config3_stage_times(args, Y, init, local_rank)
    leg['stage_ms_untimed_pass'] = stages
    if em_kernel_ms:
        step_s = leg['ms_per_step'] * 1e-3
        alg_bytes = 8.0 * U * F * T * D * args.iters
        flops = FLOPS_PER_FRAME_ITER * float(U) * F * T * args.iters
        rb = roofline_block(em_kernel_ms, U * F, args.iters, None,
                            f'; here ONE launch over {U * F} bins, three workgroups per CU')
        traffic, src = workload_pmc('config3', leg['ms_per_step'])
        rb['traffic'] = None if traffic is None else traffic['bytes_per_step']
        rb['traffic_detail'] = traffic
        rb['traffic_source'] = src
        if traffic is not None:
            rb['traffic_over_algorithmic'] = traffic['bytes_per_step'] / alg_bytes
        rb['region_ms'] = leg['ms_per_step']
        rb['whole_step'] = {
            'ms_per_step': leg['ms_per_step'],
            'fp64_valu_frac': flops / step_s / 1e12 / FP64_VALU_PEAK_TF,
            'hbm_contract_frac': alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
            'note': 'the EM kernel\'s useful flops / contract bytes over the time of the WHOLE step '
                    '(EM + DHTV alignment + PSD + gev+ban + apply of all utterances): what a rank of '
                    'an utterance-sharded run delivers; `traffic` is the PMC byte count of ALL '
                    'kernels of one step (torch copy kernels between the stages included)',
        }
        leg['roofline'] = rb
    if args.cpu_iters > 0:
        leg['cpu_baseline'] = config3_cpu_baseline(args, data, U)


def main_config3(args):
    """`--workload config3`: BASELINE configs[2] as the primary line."""
    from pb_bss_amd import _lib, engine
    world, rank, local_rank, dev, use_dist = setup(args)
    engine.set_timing(True, local_rank)
    blk, data = run_config3(args, world, rank, local_rank, dev, use_dist, primary=True)
    line = None
    if rank == 0:
        leg = next(v for k, v in blk.items() if isinstance(v, dict) and 'value' in v)
        res = {
            'metric': 'cACGMM EM iterations/sec on F=513,T=500,D=8,K=3',
            'value': leg['value'], 'unit': leg['unit'],
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': leg['ms_per_step'],
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': blk['workload'], 'utterances': blk['utterances'],
                       'em_iterations_per_step': args.iters, 'sharding': leg['sharding'],
                       'utterances_per_s': leg['utterances_per_s']},
        }
        for k in ('verify', 'comm'):
            if k in leg:
                res[k] = leg[k]
        if 'verify' in leg:
            res['mask_max_abs_err'] = leg['verify']['mask_max_abs_err']
        for k in ('roofline', 'cpu_baseline', 'stage_ms_untimed_pass'):
            if k in leg:
                res[k] = leg[k]
        line = res
    emit(line, use_dist, args)


# ------------------------------------------------------------------------------------------
# BASELINE configs[3] and configs[4]: Watson / vMF mixtures + MVDR, joint spatial + spectral model
# ------------------------------------------------------------------------------------------
C4 = dict(F=257, T=800, D=6, K=3)             # configs[3]: 6-mic CHiME-style array
C5 = dict(F=513, T=500, D=8, K=3, E=40)       # configs[4]: cACGMM x Gaussian on 40-dim embeddings


def c4_flops(D, K):
    """float64 flops per frame and EM iteration of the Watson kernel's algorithm (csrc/cwmm.hpp):
    E phase |m_k^H y|^2 as D complex multiply-adds per class (8 D + 3), log-pdf / exp / softmax
    (~35 per class); M phase Hermitian outer product P (3 D^2) and C_k += w_k P (2 D^2 K)."""
    return K * (8 * D + 38) + 3 * D * D + 2 * D * D * K


def c5_flops(D, K, E):
    """float64 flops per time-frequency point and EM iteration of the joint model: the cACGMM
    half as in the headline (2 * 3 D^2 + 4 D^2 K + ~100) plus the spherical Gaussian on the
    embedding: E-step sum_e (e - mu_k)^2 (3 E K), M-step sum w_k e (2 E K) and the shifted second
    moment (3 E + 2 K)."""
    return 2 * 3 * D * D + 4 * D * D * K + 100 + 3 * E * K + 2 * E * K + 3 * E + 2 * K


def reference_recorded(config):
    """The UNMODIFIED reference timed in the build container (tools/record_reference_timings.py
    -> profiles/reference_cpu_timings.json; /root/reference does not exist on the GPU box)."""
    path = os.path.join(ROOT, 'profiles', 'reference_cpu_timings.json')
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    cfg = rec['configs'].get(config)
    if not cfg:
        return None
    out = {k: round(v['it_per_s_median'], 3) for k, v in cfg.items() if isinstance(v, dict)
           and 'it_per_s_median' in v}
    out['unit'] = 'EM iterations/s, median of %d runs each' % next(
        v['repeats'] for v in cfg.values() if isinstance(v, dict) and 'repeats' in v)
    out['host'] = f"{rec['host']['cpu']}, {rec['host']['logical_cores']} logical cores, 1 used"
    out['source'] = 'profiles/reference_cpu_timings.json (' + rec['script'] + ')'
    return out


PORT_KIND = 'port'  # cpu_baseline.kind of the oracle-timed baselines (the contract's two values)
PORT_NOTE = ('timed in the oracle\'s reference-shaped mode (oracle/cacgmm.py REFERENCE_SHAPED: the '
             'reference\'s own einsum calls, e.g. the five-operand einsum(optimize=\'optimal\') of '
             'complex_angular_central_gaussian.py:187-196, where the restatement would be cheaper); '
             'profiles/reference_cpu_timings.json holds reference vs oracle in this mode on one host')


def median_rate(fn, iterations, repeats=3):
    """-> (median iterations/s, [runs]) of `repeats` timed calls of fn(), the oracle in its
    reference-shaped timing mode (it then costs what the reference's own calls cost)."""
    from oracle import cacgmm as oc
    runs = []
    with oc.reference_shaped():
        for _ in range(repeats):
            t1 = time.perf_counter()
            fn()
            runs.append(iterations / (time.perf_counter() - t1))
    return float(np.median(runs)), runs


SOURCES_BY_WORKLOAD = {
    'config2': KERNEL_SOURCES,
    'config3': KERNEL_SOURCES + ('dhtv.hip', 'beamform.hip'),
    'config4': ('cwmm.hpp', 'cw_inst.hip', 'cacgmm_em.hpp', 'wave_la.hpp', 'pbbss_dev.hpp',
                'em_launch.hpp', 'beamform.hip'),  # Watson leg; the vMF leg: config4_vmf
    'config4_vmf': ('vmf_bin.hip', 'embed.hip', 'embed_dev.hpp', 'pbbss_dev.hpp'),
    'config4_batched': (),  # no PMC pass of its own: the roofline block carries no traffic
    'config5': ('embed.hip', 'joint_inst.hip', 'cacgmm_em.hpp', 'wave_la.hpp', 'pbbss_dev.hpp',
                'em_launch.hpp'),
}


def workload_source_sha(workload):
    import hashlib
    h = hashlib.sha1()
    for name in SOURCES_BY_WORKLOAD[workload]:
        with open(os.path.join(ROOT, 'pb_bss_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def workload_pmc(workload, region_ms):
    """PMC traffic of one step of `workload` from a committed tools/profile_round.sh summary taken
    from exactly these kernel sources (profiles/r*_<workload>_profile.txt), or (None, reason).
    The summary's `region_trace_us` row (sum of the kernels of one step in the kernel trace) must
    not exceed this run's own region time by more than 5 %."""
    import glob
    sha = workload_source_sha(workload)
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_{workload}*_profile.txt')),
                       reverse=True):
        with open(path) as f:
            text = f.read()
        if f'kernel_source_sha: {sha}' not in text[:3000]:
            continue
        rows = {}
        for line in text.splitlines():
            parts = [x.strip() for x in line.split('|')]
            if len(parts) == 2:
                try:
                    rows[parts[0]] = float(parts[1])
                except ValueError:
                    pass
        name = 'profiles/' + os.path.basename(path)
        if 'step_fetch_bytes' not in rows:
            return None, name + ' holds no step_fetch_bytes row'
        trace_ms = rows.get('region_trace_us', 0.0) * 1e-3
        if region_ms and trace_ms > 1.05 * region_ms:
            return None, (f'{name}: kernels of one step sum to {trace_ms:.3f} ms in the trace, more '
                          f'than 5 % above this run\'s {region_ms:.3f} ms: not used as evidence')
        fetch2 = rows.get('step_fetch_bytes_x2', 2.0 * rows['step_fetch_bytes'])
        return ({'fetch_bytes_per_step_raw': rows['step_fetch_bytes'],
                 'fetch_bytes_per_step': fetch2,
                 'write_bytes_per_step': rows.get('step_write_bytes'),
                 'bytes_per_step': fetch2 + rows.get('step_write_bytes', 0.0),
                 'l2_hit_rate': rows.get('l2_hit_rate')},
                f'{name}: FETCH_SIZE (x2: the gfx950 correction of MI355X_MICROARCH.md for wide '
                f'streaming reads) + WRITE_SIZE summed over the kernels of one step, separate --pmc '
                f'passes; kernels of a step sum to {trace_ms:.3f} ms in the trace')
    return None, (f'no committed profile carries kernel_source_sha {sha} for {workload} '
                  f'(bash tools/profile_workload.sh <tag> {workload})')


def dual_roofline(region_ms, frames, iters, flops_per_frame_iter, bytes_per_iter, bytes_formula,
                  kernel, workload, primary):
    """roofline object with BOTH bounds: `primary` ('hbm' or 'fp64_valu') on top, the other one
    under `other`.  region_ms = device time of one step's EM region (HIP events in the library)."""
    sec = region_ms * 1e-3
    alg_bytes = bytes_per_iter * iters
    gbs = alg_bytes / sec / 1e9
    tf = flops_per_frame_iter * frames * iters / sec / 1e12
    traffic, src = workload_pmc(workload, region_ms)
    hbm = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': gbs / HBM_PEAK_GBS, 'algorithmic_bytes_per_step': alg_bytes,
           'algorithmic_bytes_per_iteration': bytes_per_iter, 'formula': bytes_formula}
    valu = {'bound': 'fp64_valu', 'achieved': tf, 'peak': FP64_VALU_PEAK_TF, 'unit': 'TFLOP/s',
            'frac': tf / FP64_VALU_PEAK_TF, 'flops_per_frame_iter': flops_per_frame_iter}
    top, other = (hbm, valu) if primary == 'hbm' else (valu, hbm)
    out = dict(top)
    out['other'] = other
    out['kernel'] = kernel
    out['region_ms'] = region_ms
    out['traffic'] = None if traffic is None else traffic['bytes_per_step']
    out['traffic_detail'] = traffic
    out['traffic_source'] = src
    if traffic is not None:
        out['traffic_over_algorithmic'] = traffic['bytes_per_step'] / alg_bytes
    return out


def vmf_features(Y):
    """(F, T, D) complex observation -> (F, T, 2 D) float32 unit vectors: phase-normalised to
    channel 0, real and imaginary parts stacked (what a vMF mixture clusters on an array)."""
    z = Y * np.exp(-1j * np.angle(Y[..., :1]))
    f = np.concatenate([z.real, z.imag], axis=-1).astype(np.float64)
    return (f / np.maximum(np.linalg.norm(f, axis=-1, keepdims=True), 1e-300)).astype(np.float32)


def run_config4(args, local_rank, dev, leg='watson', steps=None, warmup=None, with_cpu=True):
    """BASELINE configs[3]: mixture fit (100 EM iterations + final E-step) -> PSD -> MVDR-Souden
    with the automatic reference channel -> apply, F=257 T=800 D=6 K=3, one GPU, everything
    resident in HBM.  leg = 'watson' (CWMMTrainer's kernel on the complex observation) or 'vmf'
    (VMFMMTrainer's kernel, one mixture per bin on the 2D-dim real features)."""
    import torch
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.distribution import ComplexWatsonTrainer
    from pb_bss_amd.pipeline import device_ops as ops
    from pb_bss_amd.testing import synth
    F_, T_, D_, K_ = C4['F'], C4['T'], C4['D'], C4['K']
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    Y0, init0 = synth.make_stft(F_, T_, D_, K_, seed=0)
    y, g0 = _lib.to_device(Y0), _lib.to_device(init0)
    engine.set_timing(True, local_rank)
    if leg == 'watson':
        spline = ComplexWatsonTrainer(D_).device_spline(y.device)

        def fit():
            return engine.cwmm_fit(y, K_, spline, gamma0=g0, iterations=args.iters,
                                   final_predict=True, check_status=False)['affiliation']
    else:
        feat0 = vmf_features(Y0)
        feat = _lib.to_device(feat0)

        def fit():
            return engine.vmfmm_fit(feat, K_, gamma0=g0, iterations=args.iters,
                                    final_predict=True)['affiliation']

    def extract(masks):
        X = y.transpose(-2, -1).contiguous()                      # (F, D, T)
        psd = ops.psd(X, masks)                                   # (F, K, D, D)
        target = psd.movedim(-3, 0).contiguous()                  # (K, F, D, D)
        noise = (psd.sum(dim=-3).unsqueeze(0) - target).contiguous()
        w = ops.mvdr_souden(target, noise)                        # (K, F, D)
        return w, ops.apply_bf(w, X)                              # (K, F, T): X shared by the classes

    def step():
        masks = fit()
        if args.c4_extraction == 'off':
            return masks, None, None
        w, enh = extract(masks)
        return masks, w, enh

    read_ms = lambda lag: engine.last_kernel_ms(local_rank, lag)  # noqa: E731
    ph = preheat(step, min(args.preheat_s, 0.5), False, dev)
    elapsed, region_ms, last = timed(step, steps, warmup, False, dev, read_ms)
    ms_per_step = elapsed / steps * 1e3
    ops.assert_finite()  # the reference-channel SNR checks of the timed steps (deferred: no host
    #                      round trip inside a step since round 6)
    # the EM region alone, back to back (what the kernel trace of --workload config4 shows)
    el_fit, fit_ms, _ = timed(fit, steps, 2, False, dev, read_ms)
    name = 'CWMMTrainer (complex Watson mixture)' if leg == 'watson' else \
        'VMFMMTrainer (von-Mises-Fisher mixture, one per bin, on 2D-dim real features)'
    bytes_iter = 8.0 * F_ * T_ * D_ if leg == 'watson' else 4.0 * F_ * T_ * 2 * D_
    flops = c4_flops(D_, K_) if leg == 'watson' else (2 * 2 * D_ * K_ + 35 * K_ + 2 * 2 * D_ * K_)
    out = {
        'workload': f'BASELINE configs[3]: {name} {args.iters} EM iterations + final E-step -> PSD '
                    f'-> get_mvdr_vector_souden (automatic reference channel) per class -> apply, '
                    f'F={F_} T={T_} D={D_} K={K_}, complex64 STFT resident in HBM',
        'leg': leg,
        'value': args.iters * steps / elapsed, 'unit': 'EM iterations/s (whole chain in the step)',
        'ms_per_step': ms_per_step, 'steps': steps, 'warmup': warmup, 'dtype': 'f64',
        'em_only': {'value': args.iters * steps / el_fit, 'ms_per_fit': el_fit / steps * 1e3,
                    'region_ms': fit_ms,
                    'what': 'the fit + final E-step alone, back to back (no extraction stage)'},
        'preheat': ph,
        'roofline': dual_roofline(
            fit_ms, F_ * T_, args.iters, flops, bytes_iter,
            '8*F*T*D (one read of the complex64 observation per EM iteration, SURVEY 8d)'
            if leg == 'watson' else '4*F*T*2D (one read of the float32 features per EM iteration)',
            ('cwmm_em_wide_kernel<6,3,float> (256 bins, eight wavefronts each) + cwmm_em_split_kernel '
             '(remainder bin 256 as split groups, side stream)') if leg == 'watson' else
            'vmf_bin_em2_kernel<3,12,float,4> (one launch per fit: the whole EM loop)',
            'config4' if leg == 'watson' else 'config4_vmf', 'fp64_valu'),
    }
    # ---- the chip-filling figure: 8 utterances (2 056 bins) in ONE fit -- what a rank of an
    #      8-GPU run of a 64-utterance batch executes (the single utterance above leaves one wave
    #      per SIMD: 257 bins on 256 compute units) ----
    if args.c4_extraction == 'on':
        UB = 8
        datab = [synth.make_stft(F_, T_, D_, K_, seed=u) for u in range(UB)]
        gb = _lib.to_device(np.concatenate([d[1] for d in datab]))             # (UB F, K, T)
        if leg == 'watson':
            yb = _lib.to_device(np.concatenate([d[0] for d in datab]))         # (UB F, T, D)

            def fit_b():
                return engine.cwmm_fit(yb, K_, spline, gamma0=gb, iterations=args.iters,
                                       final_predict=True, check_status=False)['affiliation']
        else:
            fb = _lib.to_device(np.concatenate([vmf_features(d[0]) for d in datab]))

            def fit_b():
                return engine.vmfmm_fit(fb, K_, gamma0=gb, iterations=args.iters,
                                        final_predict=True)['affiliation']
        nb = max(3, steps // 2)
        el_b, fit_b_ms, last_b = timed(fit_b, nb, 2, False, dev, read_ms)
        gb_host = _lib.to_host(last_b)
        rb = dual_roofline(
            fit_b_ms, UB * F_ * T_, args.iters, flops, UB * bytes_iter,
            f'{UB} x the single-utterance figure', f'the same kernels over {UB * F_} bins in one fit',
            'config4_batched', 'fp64_valu')
        out['batched'] = {
            'utterances': UB, 'bins': UB * F_,
            'value': UB * args.iters * nb / el_b,
            'unit': 'EM iterations/s (utterance-iterations; the fit + final E-step alone)',
            'ms_per_fit': el_b / nb * 1e3, 'region_ms': fit_b_ms, 'steps': nb,
            'speedup_over_single_utterance_fit': (UB * args.iters * nb / el_b) /
                                                 (args.iters * steps / el_fit),
            'masks_finite': bool(np.isfinite(gb_host).all()),
            'first_utterance_equals_single_fit': float(np.abs(
                gb_host[:F_] - _lib.to_host(fit())).max()),
            'roofline': {k: rb[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'other',
                                            'region_ms')},
        }
    if args.check_bins and args.c4_extraction == 'on':
        from oracle import beamformer as ob
        masks, w, enh = (_lib.to_host(x) for x in last)
        fs = spread(0, F_, min(args.check_bins, 16))
        Y128 = Y0[fs].astype(np.complex128)
        if leg == 'watson':
            from oracle import cwmm as ow
            ref = ow.cwmm_predict(ow.cwmm_fit(Y128, init0[fs], iterations=args.iters), Y128)
        else:
            from oracle import embed as oe
            f64 = feat0[fs].astype(np.float64)
            ref = oe.vmfmm_predict(oe.vmfmm_fit(f64, init0[fs], args.iters), f64)
        m_err = float(np.abs(masks[fs] - ref).max())
        # extraction against the oracle fed with the DEVICE masks of ALL bins (the automatic
        # reference channel sums over every bin; the mixture itself is checked above)
        X = Y0.astype(np.complex128).transpose(0, 2, 1)
        psd = ob.psd(X, masks)
        w_err = e_err = 0.0
        for k in range(K_):
            w_ref = ob.mvdr_souden(psd[:, k], psd.sum(1) - psd[:, k])
            w_err = max(w_err, float(np.abs(w[k] - w_ref).max() / np.abs(w_ref).max()))
            s_ref = ob.apply_bf(w_ref, X)
            e_err = max(e_err, float(np.abs(enh[k] - s_ref).max() / np.abs(s_ref).max()))
        out['verify'] = {
            'bins_checked': fs, 'mask_max_abs_err': m_err, 'bf_vector_max_rel_err': w_err,
            'enhanced_max_rel_err': e_err, 'tolerance': 1e-5,
            'ok': bool(m_err < 1e-5 and w_err < 1e-5 and e_err < 1e-5),
            'includes_remainder_bin': True,
            'what': f'posterior masks after {args.iters} EM iterations vs the float64 NumPy oracle '
                    f'on evenly spaced bins (first and last included); MVDR-Souden vectors and '
                    f'enhanced signals of all classes and bins vs the oracle PSD -> mvdr_souden -> '
                    f'apply on the device masks',
        }
    if with_cpu and args.cpu_iters > 0:
        from oracle import beamformer as ob
        n = max(2, args.cpu_iters // 12)
        Y128 = Y0.astype(np.complex128)
        if leg == 'watson':
            from oracle import cwmm as ow

            def cpu():
                m = ow.cwmm_fit(Y128, init0, iterations=n)
                return ow.cwmm_predict(m, Y128)
        else:
            from oracle import embed as oe
            f64 = feat0.astype(np.float64)

            def cpu():
                return oe.vmfmm_predict(oe.vmfmm_fit(f64, init0, n), f64)

        def chain():
            mk = cpu()
            X = Y128.transpose(0, 2, 1)
            psd = ob.psd(X, mk)
            for k in range(K_):
                ob.apply_bf(ob.mvdr_souden(psd[:, k], psd.sum(1) - psd[:, k]), X)
        med, runs = median_rate(chain, n)
        out['cpu_baseline'] = {
            'value': med, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
            'runs': runs,
            'timing_mode': PORT_NOTE,
            'sample': f'NumPy oracle chain (oracle/cwmm.py / oracle/embed.py fit of {n} EM iterations '
                      f'+ predict + PSD + mvdr_souden + apply), full F={F_} T={T_} D={D_} K={K_}, '
                      f'median of 3 runs; host has {os.cpu_count()} logical cores, einsum / LAPACK '
                      f'on D x D matrices are single-threaded',
            'reference_recorded': reference_recorded('config4'),
        }
    return out


def run_config5(args, local_rank, dev, steps=None, warmup=None, with_cpu=True):
    """BASELINE configs[4] on one GPU: GCACGMMTrainer's loop (cACGMM on the STFT x spherical
    Gaussian on 40-dim embeddings, shared affiliations), 100 EM iterations + final E-step,
    everything resident in HBM."""
    from pb_bss_amd import _lib, engine
    from pb_bss_amd.testing import synth
    F_, T_, D_, K_, E_ = C5['F'], C5['T'], C5['D'], C5['K'], C5['E']
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    Y0, e0, init0 = synth.make_joint(F_, T_, D_, K_, E_, seed=0)
    y, e, g0 = _lib.to_device(Y0), _lib.to_device(e0), _lib.to_device(init0)
    engine.set_timing(True, local_rank)
    kind = _lib.EMBED_GAUSS_SPHERICAL

    def step(iters=None):
        return engine.joint_fit(y, e, K_, kind, gamma0=g0,
                                iterations=args.iters if iters is None else iters,
                                final_predict=True, check_status=False)

    read_ms = lambda lag: engine.last_kernel_ms(local_rank, lag)  # noqa: E731
    ph = preheat(step, min(args.preheat_s, 0.5), False, dev)
    elapsed, region_ms, last = timed(step, steps, warmup, False, dev, read_ms)
    ms_per_step = elapsed / steps * 1e3
    bytes_iter = 8.0 * F_ * T_ * D_ + 4.0 * F_ * T_ * E_
    out = {
        'workload': f'BASELINE configs[4] on one GPU: joint spatial + spectral model (GCACGMMTrainer: '
                    f'cACGMM on the STFT x spherical Gaussian on {E_}-dim float32 embeddings, shared '
                    f'affiliations), {args.iters} EM iterations + final E-step, F={F_} T={T_} D={D_} '
                    f'K={K_}, resident in HBM',
        'value': args.iters * steps / elapsed, 'unit': 'EM iterations/s',
        'ms_per_step': ms_per_step, 'us_per_iteration': ms_per_step * 1e3 / args.iters,
        'steps': steps, 'warmup': warmup, 'dtype': 'f64', 'preheat': ph,
        'status_bits_or': int(np.bitwise_or.reduce(_lib.to_host(last['status']).ravel())),
        'roofline': dual_roofline(
            region_ms, F_ * T_, args.iters, c5_flops(D_, K_, E_), bytes_iter,
            '8*F*T*D + 4*F*T*E (one read of the complex64 observation and of the float32 '
            'embedding per EM iteration)',
            'per iteration: embedding E-step + cacgmm_joint_kernel<8,3> + embedding M-step sweep + '
            'finalize (region_ms = HIP events around the whole enqueued loop, inside the library)',
            'config5', 'hbm'),
    }
    if args.check_bins:
        # the spectral mixture couples every bin, so the oracle runs the FULL problem -- over the
        # same number of iterations the timed steps ran (0.3 s per iteration on a host core); the
        # same run is the CPU baseline's sample.  A 4-iteration check rides along: it separates
        # "wrong from the start" from "drifts over the trajectory" should the long one ever fail.
        from oracle import embed as oe
        Y128, e64 = Y0.astype(np.complex128), e0.astype(np.float64)
        n_short = 4
        got = _lib.to_host(step(n_short)['affiliation'])
        ref = oe.joint_model_predict(oe.joint_fit('gaussian', Y128, e64, init0, n_short), Y128, e64)
        err_short = float(np.abs(got - ref).max())
        n = args.iters if (with_cpu and args.cpu_iters > 0) else n_short
        g100 = _lib.to_host(last['affiliation'])
        err, cpu_s = err_short, None
        if n != n_short:
            from oracle import cacgmm as oc
            t1 = time.perf_counter()
            with oc.reference_shaped():  # same results up to rounding; costs what the reference costs
                ref_model = oe.joint_fit('gaussian', Y128, e64, init0, n)
            cpu_s = time.perf_counter() - t1
            err = float(np.abs(g100 - oe.joint_model_predict(ref_model, Y128, e64)).max())
        out['verify'] = {
            'mask_max_abs_err': err, 'iterations_checked': n, 'tolerance': 1e-6,
            'mask_max_abs_err_after_4_iterations': err_short,
            'bins_checked': F_, 'includes_remainder_bin': True,
            'ok': bool(err < 1e-6 and err_short < 1e-6 and np.isfinite(g100).all()
                       and abs(float(g100.sum(1).mean()) - 1.0) < 1e-9),
            'what': f'posterior masks of ALL {F_} bins of the timed steps ({n} EM iterations + final '
                    f'E-step) vs the float64 NumPy oracle run over the same {n} iterations '
                    f'(oracle/embed.py joint_fit; the spectral mixture couples the bins, so the '
                    f'oracle runs the full problem), and after {n_short} iterations',
        }
        if cpu_s is not None:
            out['cpu_baseline'] = {
                'value': n / cpu_s, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
                'runs': [n / cpu_s],
                'timing_mode': PORT_NOTE,
                'sample': f'NumPy oracle joint_fit (oracle/embed.py), full F={F_} T={T_} D={D_} '
                          f'K={K_} E={E_}, ONE run of {n} EM iterations ({cpu_s:.1f} s; the run the '
                          f'verify block compares against); host has {os.cpu_count()} logical cores, '
                          f'1 used',
                'reference_recorded': reference_recorded('config5'),
            }
    return out


def main_config45(args):
    """`--workload config4|config5`: that configuration as the primary line (profiling runs)."""
    world, rank, local_rank, dev, use_dist = setup(args)
    assert world == 1, 'configs[3] / [4] are measured on one GPU'
    if args.workload == 'config4':
        blk = run_config4(args, local_rank, dev, leg=args.leg)
        metric = 'mixture-model EM iterations/sec on F=257,T=800,D=6,K=3 (+ MVDR-Souden)'
    else:
        blk = run_config5(args, local_rank, dev)
        metric = 'joint GCACGMM EM iterations/sec on F=513,T=500,D=8,K=3,E=40'
    res = {'metric': metric, 'value': blk['value'], 'unit': blk['unit'], 'n_gpus': 1,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': blk['ms_per_step'],
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
           'data': 'synthetic', 'config': {'workload': blk['workload']}}
    res.update({k: v for k, v in blk.items() if k not in ('value', 'unit', 'ms_per_step', 'workload',
                                                         'steps', 'warmup', 'dtype')})
    emit(res, use_dist, args)


def has_f32():
    from pb_bss_amd import _lib
    return _lib.load().pbbss_version() >= 300




# ------------------------------------------------------------------------------------------
# parity of the headline, summaries for the compact line
# ------------------------------------------------------------------------------------------
def verify_headline(out, args, world, data, got_all, got2, same, precision):
    """Parity of the headline run (in place: out['verify'], out['mask_max_abs_err']): the gathered
    masks `got_all` (world, F, K, T) -- `got2` for the packed-FP32 kernel -- against the oracle on
    bins of EVERY rank's shard of EVERY utterance.  The oracle is the checker, never timed here."""
    from pb_bss_amd.sharding import shard_bounds
    from oracle import cacgmm as oc
    nb = args.check_bins if world == 1 else max(3, args.check_bins // (world * world))
    worst, per_shard, nchk = 0.0, [0.0] * world, 0
    # single precision cannot be compared over a 100-iteration trajectory (the
    # reference's own float32 path drifts 2e-2 from its float64 path): the packed kernel is
    # checked per step -- two iterations from the same initialisation -- at the parity tests'
    # tolerance
    tol, n_it = (1e-5, args.iters) if precision == 'f64' else (2e-4, 2)
    for u in range(world):
        sel = []
        for rr in range(world):
            slo, shi = shard_bounds(F, world, rr)
            sel += [(rr, f) for f in spread(slo, shi, nb)]
        fs = [f for _, f in sel]
        Y128 = data[u][0][fs].astype(np.complex128)
        ref = oc.em_predict(oc.em_fit(Y128, data[u][1][fs], iterations=n_it), Y128)
        got_u = got_all[u] if precision == 'f64' else got2[u]
        for j, (rr, f) in enumerate(sel):
            e = float(np.abs(got_u[f] - ref[j]).max())
            per_shard[rr] = max(per_shard[rr], e)
            worst = max(worst, e)
        nchk += len(sel)
    out['mask_max_abs_err'] = worst
    out['verify'] = {
        'bins_checked': nchk, 'bins_per_shard_per_utterance': nb, 'utterances_checked': world,
        'max_abs_err_per_rank_shard': per_shard, 'tolerance': tol,
        'ok': bool(worst < tol),
        'includes_remainder_bin': True,
        'gathered_masks_identical_on_all_ranks': same,
        'what': ('posterior masks after all EM iterations' if precision == 'f64' else
                 'posterior masks after TWO EM iterations from the same initialisation (per-step '
                 'check of the single-precision kernel; separate untimed launch)') +
                ' vs the float64 NumPy oracle on bins drawn from every rank\'s shard (first, '
                'last and evenly spaced bins) of every utterance; checksum of the gathered '
                'tensor compared across ranks',
    }


def strict(o):
    """Non-finite floats -> None (json.dumps(allow_nan=False) would raise on them; a bare NaN
    makes the line unparsable); numpy scalars -> Python numbers; tuples -> lists."""
    if isinstance(o, dict):
        return {str(k): strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [strict(v) for v in o]
    if isinstance(o, (bool, np.bool_)):
        return bool(o)
    if isinstance(o, (int, np.integer)):
        return int(o)
    if isinstance(o, (float, np.floating)):
        return float(o) if math.isfinite(float(o)) else None
    return o


def _sig(x, n=6):
    return float(f'{x:.{n}g}') if isinstance(x, float) and math.isfinite(x) else x


def _get(o, *path):
    for p in path:
        if not isinstance(o, dict) or p not in o:
            return None
        o = o[p]
    return o


def _summary(blk, frac_of=None):
    """One-number summary of a workload block: value, ms_per_step, roofline fraction(s), traffic
    ratio, parity.  The full block is in the side file."""
    if not isinstance(blk, dict):
        return None
    if 'error' in blk and 'value' not in blk:   # a block bench.py's guard recorded as failed
        return {'error': str(blk['error'])[:160]}
    rf = blk.get('roofline') or {}
    other = rf.get('other') or rf.get('hbm_contract') or {}
    v = blk.get('verify') or {}
    err = v.get('mask_max_abs_err')
    if err is None and v.get('max_abs_err_per_rank_shard'):
        err = max(v['max_abs_err_per_rank_shard'])
    s = {'value': blk.get('value'), 'ms_per_step': blk.get('ms_per_step'),
         'bound': rf.get('bound'), 'frac': rf.get('frac'),
         ('frac_' + str(other.get('bound'))): other.get('frac'),
         'traffic_over_algorithmic': rf.get('traffic_over_algorithmic'),
         'mask_err': err, 'tolerance': v.get('tolerance'), 'ok': v.get('ok'),
         'cpu_baseline': _get(blk, 'cpu_baseline', 'value')}
    for k in ('em_only', 'batched'):
        if isinstance(blk.get(k), dict):
            s[k] = {n: _sig(x) for n, x in (('value', blk[k].get('value')),
                                            ('frac', _get(blk[k], 'roofline', 'frac')))
                    if x is not None}
    return {k: _sig(x) if isinstance(x, float) else x for k, x in s.items()
            if x is not None and k != 'frac_None'}


def strong_block(c3, world):
    """Top-level `strong`: BASELINE configs[2] -- a FIXED batch of 64 utterances through EM -> mask
    all-gather -> DHTV -> PSD -> gev+ban -> apply -- on this run's N GPUs: the strong-scaling
    curve of the metric, readable without nested blocks."""
    if not isinstance(c3, dict):
        return None
    legs = {k: v for k, v in c3.items() if isinstance(v, dict) and 'value' in v}
    if not legs:
        return None
    name, best = max(legs.items(), key=lambda kv: kv[1]['value'])
    n1 = recorded_n1_config3_ms()
    out = {'workload': f'BASELINE configs[2]: {c3.get("utterances")} utterances x 100 EM iterations '
                       '+ DHTV + PSD + gev+ban + apply, fixed batch (strong scaling)',
           'n_gpus': world, 'sharding': name, 'value': best['value'],
           'unit': 'EM iterations/s (utterance-iterations, whole chain)',
           'ms_per_step': best['ms_per_step'], 'ok': _get(best, 'verify', 'ok'),
           'mask_err': _get(best, 'verify', 'mask_max_abs_err'),
           'per_sharding_ms_per_step': {k: v['ms_per_step'] for k, v in legs.items()}}
    if n1:
        out['n1_ms_per_step_recorded'] = n1['ms_per_step']
        out['n1_source'] = n1['source']
        out['speedup_vs_recorded_n1'] = n1['ms_per_step'] / best['ms_per_step']
    return {k: _sig(v) if isinstance(v, float) else v for k, v in out.items()}


def recorded_n1_config3_ms():
    """ms per step of BASELINE configs[2] on ONE MI355X from the newest committed bench line
    (profiles/r*_bench.json, N = 1, 64 utterances): the denominator of `strong.speedup_vs_recorded_n1`
    when this run has N > 1 (the N = 1 run of the same sweep is the driver's to compare)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench*.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get('n_gpus') != 1:
            continue
        ms = None
        c3 = d.get('config3') or {}
        if isinstance(c3.get('single'), dict) and c3.get('utterances', 64) == 64:
            ms = c3['single'].get('ms_per_step')
        st = d.get('strong') or {}
        if ms is None and st.get('n_gpus') == 1:
            ms = st.get('ms_per_step')
        if ms:
            return {'ms_per_step': ms, 'source': 'profiles/' + os.path.basename(path)}
    return None
