"""CPU, world_size 2 over gloo: the N>1 data path of pb_bss_amd.sharding --
uneven contiguous bin blocks, pad / all-gather / trim -- reproduces the full
tensor on every rank.  (On GPUs the same code runs over RCCL.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pb_bss_amd.sharding import all_gather_bins, shard_bounds, shard_sizes


def test_shard_bounds_cover_all_bins():
    for F in (1, 7, 129, 257, 513):
        for world in (1, 2, 3, 4, 8):
            sizes = shard_sizes(F, world)
            assert sum(sizes) == F and max(sizes) - min(sizes) <= 1
            edges = [shard_bounds(F, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == F
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
    assert shard_sizes(513, 8) == [65] + [64] * 7


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, F, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        full = torch.from_numpy(rng.uniform(size=(3, F, 2, 5)))  # (utt, F, K, T)
        lo, hi = shard_bounds(F, world, rank)
        got = all_gather_bins(full[:, lo:hi].contiguous(), F, bin_axis=1)
        ok = got.shape == full.shape and bool((got == full).all())
        got0 = all_gather_bins(full[0, lo:hi].contiguous(), F, bin_axis=0)
        ok = ok and bool((got0 == full[0]).all())
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('F', [513, 8, 5])
def test_all_gather_bins_world2(F):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), F, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------
# The config-3 chain (pb_bss_amd.pipeline.separate) under gloo: the REAL orchestration code --
# bin / utterance slicing, mask all-gather (also in float32), per-rank DHTV shares, mapping
# gather, per-bin extraction, output gather -- with the NumPy oracle standing in for the five
# device stages.  Sharded runs must reproduce the single-process run.
class _oracle_ops:
    """pb_bss_amd.pipeline.device_ops with oracle bodies (CPU torch tensors in and out)."""

    @staticmethod
    def em_masks(Y, init, iterations):
        from oracle import cacgmm as oc
        out = []
        for y, g in zip(Y.numpy(), init.numpy()):
            y128 = y.astype(np.complex128)
            out.append(oc.em_predict(oc.em_fit(y128, g, iterations=iterations), y128))
        return torch.from_numpy(np.stack(out))

    @staticmethod
    def dhtv_mapping(mask_kft, stft_size):
        from oracle import permutation_alignment as op
        plan = op.alignment_plan(stft_size, **op.PRESETS[stft_size])
        return torch.from_numpy(np.stack([op.dhtv_calculate_mapping(m, plan)
                                          for m in mask_kft.numpy()]))

    @staticmethod
    def apply_mapping(mask_kft, mapping):
        from oracle import permutation_alignment as op
        return torch.from_numpy(np.stack([op.apply_mapping(m, p)
                                          for m, p in zip(mask_kft.numpy(), mapping.numpy())]))

    @staticmethod
    def psd(X, mask_fkt):
        from oracle import beamformer as ob
        return torch.from_numpy(np.stack([ob.psd(x.astype(np.complex128), m)
                                          for x, m in zip(X.numpy(), mask_fkt.numpy())]))

    @staticmethod
    def gev_ban(target, noise):
        from oracle import beamformer as ob
        t, n = target.numpy(), noise.numpy()
        D = t.shape[-1]
        w = ob.bf_vector('gev+ban', t.reshape(-1, D, D), n.reshape(-1, D, D))
        # GEV vectors carry an arbitrary phase: fix it (first sensor real, positive) so that
        # sharded and unsharded runs can be compared entry by entry
        w = w * np.exp(-1j * np.angle(w[..., :1]))
        return torch.from_numpy(w.reshape(t.shape[:-1]))

    @staticmethod
    def apply_bf(w, X):
        from oracle import beamformer as ob
        return torch.from_numpy(ob.apply_bf(w.numpy(), X.numpy().astype(np.complex128)))


def _pipeline_inputs(U=3, F=257, T=40, D=3, K=2):
    from pb_bss_amd.testing import synth
    data = [synth.make_stft(F, T, D, K, seed=70 + u) for u in range(U)]
    return (torch.from_numpy(np.stack([d[0] for d in data])),
            torch.from_numpy(np.stack([d[1] for d in data])))


def _pipeline_worker(rank, world, port, shard, gather_dtype, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pb_bss_amd import pipeline
        Y, init = _pipeline_inputs()
        ref = pipeline.separate(Y, init, 3, 512, ops=_oracle_ops)
        got = pipeline.separate(Y, init, 3, 512, shard=shard, gather_output=True,
                                mask_gather_dtype=gather_dtype, ops=_oracle_ops)
        ok = all(got[k].shape == ref[k].shape for k in ref)
        ok = ok and bool((got['mapping'] == ref['mapping']).all())
        for k in ('masks', 'enhanced', 'bf_vector'):
            ok = ok and float((got[k] - ref[k]).abs().max()) < 1e-12
        # without the output gather every rank holds exactly its own block
        loc = pipeline.separate(Y, init, 3, 512, shard=shard, mask_gather_dtype=gather_dtype,
                                ops=_oracle_ops)
        if shard == 'bins':
            lo, hi = shard_bounds(Y.shape[1], world, rank)
            ok = ok and float((loc['enhanced'] - ref['enhanced'][:, :, lo:hi]).abs().max()) < 1e-12
        else:
            lo, hi = shard_bounds(Y.shape[0], world, rank)
            ok = ok and float((loc['enhanced'] - ref['enhanced'][lo:hi]).abs().max()) < 1e-12
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('shard,gather_dtype', [('bins', None), ('bins', torch.float32),
                                                ('utterances', None)])
def test_config3_chain_sharded_equals_single_process(shard, gather_dtype):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pipeline_worker, args=(world, _free_port(), shard, gather_dtype, ret),
             nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------
# weight_constant_axis containing the SHARDED bin axis: the all-reduce hook between the E- and
# the M-step (sharding.shared_weight_allreduce) must reproduce estimate_mixture_weight on the
# full array, with and without a saliency, on every rank (CPU tensors over gloo here; the same
# code all-reduces device tensors over RCCL).
def _hook_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle.cacgmm import estimate_mixture_weight  # the checker (CPU box: no device)
        from pb_bss_amd.sharding import shared_weight_allreduce
        rng = np.random.default_rng(3)
        U, F, K, T = 2, 7, 3, 11
        aff = rng.uniform(size=(U, F, K, T))
        aff /= aff.sum(axis=-2, keepdims=True)
        sal = rng.uniform(size=(U, F, T))
        sal[0, :, 4] = 0.0  # a frame that is masked out in every bin: the `where(norm == 0)` branch
        lo, hi = shard_bounds(F, world, rank)
        ok = True
        for axis in ((-3,), (-3, -1), [-1, -3]):
            for with_sal in (False, True):
                hook = shared_weight_allreduce(axis, F, bin_axis=-3)
                got = hook(torch.from_numpy(aff[:, lo:hi].copy()),
                           torch.from_numpy(sal[:, lo:hi].copy()) if with_sal else None)
                ref = estimate_mixture_weight(aff, sal if with_sal else None,
                                              weight_constant_axis=tuple(axis))
                ok = ok and got.shape == ref.shape and bool(np.abs(got.numpy() - ref).max() < 1e-14)
        # a rank without bins keeps the collective schedule
        hook = shared_weight_allreduce((-3,), F, bin_axis=-3)
        if rank == 0:
            w = hook(torch.from_numpy(aff.copy()), None)
            ok = ok and bool(np.abs(w.numpy() - aff.mean(axis=-3, keepdims=True)).max() < 1e-14)
        else:
            hook.idle((U, 1, K, T), torch.float64, 'cpu', 1)
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_shared_weight_allreduce_hook_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hook_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


class _OracleDHTVAligner:
    """NumPy stand-in of a foreign aligner object: the oracle's DHTV solver."""

    def __init__(self, stft_size):
        from oracle import permutation_alignment as op
        self.plan = op.alignment_plan(stft_size, **op.PRESETS[stft_size])

    def calculate_mapping(self, mask_kft):
        from oracle import permutation_alignment as op
        return op.dhtv_calculate_mapping(np.asarray(mask_kft), self.plan)

    def apply_mapping(self, mask_kft, mapping):
        from oracle import permutation_alignment as op
        return op.apply_mapping(np.asarray(mask_kft), np.asarray(mapping))


def _aligner_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pb_bss_amd.sharding import sharded_inline_aligner
        from pb_bss_amd.distribution.mixture_model_utils import apply_inline_permutation_alignment
        rng = np.random.default_rng(11)
        K, F, T = 3, 257, 60
        act = rng.uniform(size=(K, 1, T)) ** 4
        mask = act * rng.uniform(0.5, 1.0, size=(K, F, T)) + 0.05 * rng.uniform(size=(K, F, T))
        mask /= mask.sum(0, keepdims=True)
        for f in range(F):
            mask[:, f] = mask[rng.permutation(K), f]
        q = rng.uniform(1, 2, size=(F, K, T))
        inner = _OracleDHTVAligner(512)
        want_map = inner.calculate_mapping(mask)
        want = inner.apply_mapping(mask, want_map)
        lo, hi = shard_bounds(F, world, rank)
        al = sharded_inline_aligner(inner, F)
        ok = True
        # NumPy blocks (a foreign aligner on host arrays) and CPU tensors alike
        for wrap in (lambda x: x, torch.from_numpy):
            local = wrap(np.ascontiguousarray(mask[:, lo:hi]))
            m = al.calculate_mapping(local)
            ok = ok and np.array_equal(np.asarray(m), want_map[:, lo:hi])
            got = al.apply_mapping(local, m)
            ok = ok and np.array_equal(np.asarray(got), want[:, lo:hi])
        # through the trainer-side helper, (F, K, T) blocks with the quadratic form riding along
        aff_loc = np.ascontiguousarray(mask[:, lo:hi].transpose(1, 0, 2))
        a2, q2 = apply_inline_permutation_alignment(
            affiliation=aff_loc, quadratic_form=q[lo:hi], weight_constant_axis=(-3,), aligner=al)
        ok = ok and np.array_equal(a2, want[:, lo:hi].transpose(1, 0, 2))
        ok = ok and np.array_equal(
            q2, inner.apply_mapping(q.transpose(1, 0, 2), want_map)[:, lo:hi].transpose(1, 0, 2))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_inline_aligner_world2():
    """inline_permutation_aligner under bin sharding: the gathered masks give every rank the
    mapping an unsharded run computes; each applies the columns of its own block"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_aligner_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------
# MVDR-Souden with the automatic reference channel under bin sharding: the SNR of
# beamformer.py:616-620 sums over ALL bins, so the per-problem sums are all-reduced
# (pipeline.separate(beamformer='mvdr_souden'), extraction.get_mvdr_vector_souden(shard_group=)).
class _oracle_ops_souden(_oracle_ops):
    @staticmethod
    def mvdr_souden(target, noise, shard_group=None):
        """Same contract as device_ops.mvdr_souden, oracle arithmetic: Phi_nn^-1 Phi_xx / tr per
        bin (beamformer.py:627-698), the reference channel from the (all-reduced) SNR sums."""
        from oracle import beamformer as ob
        from pb_bss_amd.extraction.beamformer import _select_reference_channel_sharded
        from pb_bss_amd.pipeline import select_column
        t, n = target.numpy(), np.broadcast_to(noise.numpy(), target.shape)
        eps = np.finfo(np.float64).tiny
        phi = ob.stable_solve(n, t)
        lam = np.trace(phi, axis1=-1, axis2=-2)[..., None, None]
        mat = phi / np.maximum(lam.real, eps)
        num = np.einsum('...fdr,...fde,...fer->...fr', mat.conj(), t, mat)
        den = np.einsum('...fdr,...fde,...fer->...fr', mat.conj(), n, mat)
        ref = _select_reference_channel_sharded(
            torch.from_numpy(num), torch.from_numpy(den), eps,
            shard_group if shard_group is not None else None) if shard_group is not None else \
            np.argmax((num.sum(-2) / np.maximum(den.sum(-2), eps)).real, axis=-1)
        return select_column(torch.from_numpy(mat), ref)


def _souden_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import beamformer as ob
        from pb_bss_amd import pipeline
        Y, init = _pipeline_inputs(U=2, F=257, T=30, D=3, K=2)
        ref = pipeline.separate(Y, init, 3, 512, ops=_oracle_ops_souden, beamformer='mvdr_souden')
        ok = True
        for shard in ('bins', 'utterances', 'auto'):  # auto: two utterances, two ranks -> utterances
            got = pipeline.separate(Y, init, 3, 512, shard=shard, gather_output=True,
                                    ops=_oracle_ops_souden, beamformer='mvdr_souden')
            ok = ok and bool((got['mapping'] == ref['mapping']).all())
            for k in ('masks', 'enhanced', 'bf_vector'):
                ok = ok and float((got[k] - ref[k]).abs().max()) < 1e-12
        # shard='auto' with ONE utterance on two ranks: only the bins can be split
        ref1 = pipeline.separate(Y[0], init[0], 3, 512, ops=_oracle_ops_souden,
                                 beamformer='mvdr_souden')
        got1 = pipeline.separate(Y[0], init[0], 3, 512, shard='auto', gather_output=True,
                                 ops=_oracle_ops_souden, beamformer='mvdr_souden')
        ok = ok and bool((got1['mapping'] == ref1['mapping']).all())
        for k in ('masks', 'enhanced', 'bf_vector'):
            ok = ok and float((got1[k] - ref1[k]).abs().max()) < 1e-12
        # ... and the unsharded stand-in is the reference function itself, problem by problem
        X = Y.numpy().astype(np.complex128).transpose(0, 1, 3, 2)
        aligned = ref['masks'].numpy()                                   # (U, K, F, T)
        for u in range(X.shape[0]):
            psd = ob.psd(X[u], aligned[u].transpose(1, 0, 2))            # (F, K, D, D)
            for k in range(psd.shape[1]):
                w = ob.mvdr_souden(psd[:, k], psd.sum(1) - psd[:, k])
                ok = ok and float(np.abs(w - ref['bf_vector'][u, k].numpy()).max()) < 1e-10
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_mvdr_souden_reference_channel_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_souden_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------
# sharding.fit_predict_sharded with trainers other than CACGMMTrainer (BASELINE configs[3]:
# complex-Watson / vMF mixtures): the REAL function -- slicing of y / initialization / saliency,
# the empty-shard branch, the gather -- with oracle-backed stand-ins for the device trainers.
class _OracleCWMMTrainer:
    _to_device = staticmethod(lambda x: x)  # CPU stand-in: tensors stay where they are

    def fit_predict(self, y, initialization, iterations, saliency=None,
                    weight_constant_axis=(-1,)):
        from oracle import cwmm as ow
        y128 = y.numpy().astype(np.complex128)
        m = ow.cwmm_fit(y128, initialization.numpy(), iterations=iterations,
                        saliency=None if saliency is None else saliency.numpy(),
                        weight_constant_axis=weight_constant_axis, spline_markers=200)
        return torch.from_numpy(ow.cwmm_predict(m, y128))


class _OracleVMFMMTrainer:
    _to_device = staticmethod(lambda x: x)

    def fit_predict(self, y, initialization, iterations, saliency=None,
                    weight_constant_axis=(-1,)):
        from oracle import embed as oe
        m = oe.vmfmm_fit(y.numpy(), initialization.numpy(), iterations=iterations,
                         saliency=None if saliency is None else saliency.numpy(),
                         weight_constant_axis=weight_constant_axis)
        return torch.from_numpy(oe.vmfmm_predict(m, y.numpy()))


def _trainer_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pb_bss_amd.sharding import fit_predict_sharded
        from pb_bss_amd.testing import synth
        ok = True
        # Watson mixture: bins are independent problems (cwmm.py:76-149)
        Y, init = synth.make_stft(9, 50, 4, 2, seed=5)
        sal = np.random.default_rng(1).uniform(0.5, 1.0, size=(9, 50))
        for kw in ({}, {'saliency': torch.from_numpy(sal)}):
            full = _OracleCWMMTrainer().fit_predict(torch.from_numpy(Y), torch.from_numpy(init), 3, **kw)
            got = fit_predict_sharded(torch.from_numpy(Y), torch.from_numpy(init), 3,
                                      trainer=_OracleCWMMTrainer(), **kw)
            ok = ok and got.shape == full.shape and float((got - full).abs().max()) < 1e-12
        # more ranks than bins: rank 1 owns nothing
        got = fit_predict_sharded(torch.from_numpy(Y[:1]), torch.from_numpy(init[:1]), 2,
                                  trainer=_OracleCWMMTrainer)
        full = _OracleCWMMTrainer().fit_predict(torch.from_numpy(Y[:1]), torch.from_numpy(init[:1]), 2)
        ok = ok and float((got - full).abs().max()) < 1e-12
        # vMF mixture with an independent leading axis (vmfmm.py:124-172): (B, N, E), (B, K, N)
        rng = np.random.default_rng(2)
        emb = rng.standard_normal((5, 40, 6))
        ini = rng.uniform(size=(5, 3, 40))
        ini /= ini.sum(axis=1, keepdims=True)
        full = _OracleVMFMMTrainer().fit_predict(torch.from_numpy(emb), torch.from_numpy(ini), 3)
        got = fit_predict_sharded(torch.from_numpy(emb), torch.from_numpy(ini), 3,
                                  trainer=_OracleVMFMMTrainer(), bin_axis=0)
        ok = ok and float((got - full).abs().max()) < 1e-12
        # weights shared over the sharded axis need a collective only CACGMMTrainer has
        try:
            fit_predict_sharded(torch.from_numpy(Y), torch.from_numpy(init), 2,
                                trainer=_OracleCWMMTrainer(), weight_constant_axis=(-3,))
            ok = False
        except NotImplementedError:
            pass
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_fit_predict_sharded_other_trainers_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------
# Joint spatial + spectral models under bin sharding (BASELINE configs[4]): the spectral half
# is ONE mixture over all F*T points, so its M-step sums are all-reduced once per EM iteration
# (in the library: ncclAllReduce between the partial-sum kernel and the finalize kernel,
# pbbss_mix_opts.sharded).  This test pins the DECOMPOSITION the kernels rely on -- local
# weighted sums (S0, S1, then S2 about the global mean), one all-reduce each, the reference's
# formulas on the totals -- against the oracle's unsharded joint fit, over gloo.
def _joint_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import cacgmm as oc, embed as oe
        from pb_bss_amd.sharding import all_reduce_sum
        from pb_bss_amd.testing import synth
        F, T, D, K, E = 7, 40, 3, 2, 5
        Y, emb, init = synth.make_joint(F, T, D, K, E, seed=3, embedding_dtype=np.float64)
        Y = Y.astype(np.complex128)
        ref = oe.joint_fit('gaussian', Y, emb, init, iterations=4)
        ref_aff = oe.joint_model_predict(ref, Y, emb)
        lo, hi = shard_bounds(F, world, rank)
        yn, e_loc = oe.unit_rows(Y[lo:hi]), emb[lo:hi]
        aff, q, model = init[lo:hi], np.ones_like(init[lo:hi]), None
        for _ in range(4):
            if model is not None:
                aff, q = oe.joint_predict(model, yn, e_loc, 1e-10, False)
            masked = aff                                                  # saliency = 1
            flat = e_loc.reshape(-1, E)
            mk = masked.transpose(1, 0, 2).reshape(K, -1)
            s01 = torch.from_numpy(np.concatenate([mk.sum(-1, keepdims=True), mk @ flat], axis=1))
            s01 = all_reduce_sum(s01).numpy()                             # (K, 1 + E) totals
            mean = s01[:, 1:] / np.maximum(s01[:, :1], np.finfo(np.float64).tiny)
            s2 = torch.from_numpy(np.einsum('kn,kne->k', mk, (flat[None] - mean[:, None]) ** 2))
            s2 = all_reduce_sum(s2).numpy()
            cov = s2 / (np.maximum(s01[:, 0], np.finfo(np.float64).tiny) * E)
            eigvec, eigval = oc.cacg_m_step(np.swapaxes(yn[..., None, :, :], -1, -2), masked, q,
                                            eigenvalue_floor=1e-10)
            model = dict(kind='gaussian', weight=oe.joint_weight(masked, (-1,)),
                         weight_constant_axis=(-1,), spatial_weight=1., spectral_weight=1.,
                         mean=mean, covariance=cov, covariance_type='spherical',
                         eigvec=eigvec, eigval=eigval)
        got = oe.joint_predict(model, yn, e_loc)[0]
        ok = float(np.abs(got - ref_aff[lo:hi]).max()) < 1e-10
        ok = ok and float(np.abs(model['mean'] - ref['mean']).max()) < 1e-12
        ok = ok and float(np.abs(model['covariance'] - ref['covariance']).max()) < 1e-12
        full = all_gather_bins(torch.from_numpy(got), F, bin_axis=0).numpy()
        ok = ok and float(np.abs(full - ref_aff).max()) < 1e-10
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_joint_model_spectral_sums_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_joint_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------
# A rank WITHOUT bins (world > F) next to an inline aligner and bin-constant weights: the idle
# rank must issue its collectives in the order of the step-wise loop of the ranks that own bins
# (cacgmm.py:252-278: weight all-reduce in iteration 0, then mask gather + weight all-reduce per
# iteration) -- gloo pairs collectives by sequence, a different order mismatches or hangs.
def _stepwise_cpu_trainer():
    from pb_bss_amd.distribution import CACGMMTrainer

    class _StepwiseCPUTrainer(CACGMMTrainer):
        """CPU stand-in with the collective schedule of CACGMMTrainer's step-wise loop; the
        arithmetic is a toy (the schedule is what is under test)."""
        _to_device = staticmethod(lambda x: x if isinstance(x, torch.Tensor) else torch.from_numpy(x))

        def fit_predict(self, y, initialization=None, iterations=100, *, weight_constant_axis=(-1,),
                        inline_permutation_aligner=None, _weight_hook=None, **_):
            aff = initialization.clone()                         # (F_local, K, T)
            weights = []
            for it in range(iterations):
                if it > 0:
                    aff = aff * (1.0 + 0.1 * it)
                    aff = aff / aff.sum(-2, keepdim=True)
                    kft = aff.transpose(0, 1).contiguous()
                    mapping = inline_permutation_aligner.calculate_mapping(kft)
                    aff = inline_permutation_aligner.apply_mapping(kft, mapping).transpose(0, 1)
                weights.append(_weight_hook(aff, None))
            self.weights = weights
            return aff.contiguous()

    return _StepwiseCPUTrainer()


class _IdentityAligner:
    def calculate_mapping(self, kft):
        K, F, _ = kft.shape
        return np.tile(np.arange(K)[:, None], (1, F))

    def apply_mapping(self, kft, mapping):
        return kft


def _idle_order_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pb_bss_amd.sharding import fit_predict_sharded
        rng = np.random.default_rng(5)
        F, K, T, D = 1, 2, 12, 3                                 # world 2 > F: rank 1 owns nothing
        y = torch.from_numpy(rng.normal(size=(F, T, D)) + 1j * rng.normal(size=(F, T, D)))
        init = rng.uniform(size=(F, K, T))
        init = torch.from_numpy(init / init.sum(1, keepdims=True))
        tr = _stepwise_cpu_trainer()
        got = fit_predict_sharded(y, init, iterations=4, trainer=tr, weight_constant_axis=(-3,),
                                  inline_permutation_aligner=_IdentityAligner())
        # single-process expectation of the toy loop
        aff = init.clone()
        for it in range(1, 4):
            aff = aff * (1.0 + 0.1 * it)
            aff = aff / aff.sum(-2, keepdim=True)
        ok = got.shape == (F, K, T) and torch.allclose(got, aff, atol=1e-14)
        if rank == 0:  # the owner's weights are means over ALL (= its own) bins
            ok = ok and len(tr.weights) == 4 and torch.allclose(
                tr.weights[-1], aff.mean(dim=0, keepdim=True), atol=1e-14)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_idle_rank_keeps_collective_order_with_aligner_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_idle_order_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
