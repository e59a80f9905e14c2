"""GPU: the multi-GPU exchange step behind the C ABI (csrc/comm.hip).

One GPU is available to the test suite, so the RCCL collective itself runs with world_size 1
(through pbbss_comm_create / pbbss_allgather_masks), while the trimming logic for uneven shards
-- the part that differs between ranks -- is driven with synthetic gathered buffers for
world_size 2, 3 and 8 through pbbss_allgather_unpack.  The same orchestration with
torch.distributed is covered on CPU by tests/test_sharding_gloo.py (world_size 2, gloo)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world', [1, 2, 3, 8])
@pytest.mark.parametrize('dtype', ['float64', 'float32', 'int64'])
def test_allgather_unpack_trims_uneven_shards(world, dtype):
    import torch
    from pb_bss_amd import _lib
    from pb_bss_amd.sharding import shard_bounds
    rng = np.random.default_rng(world)
    outer, F, inner = 3, 13, 10
    full = (rng.standard_normal((outer, F, inner)) * 1000).astype(dtype)
    pad = -(-F // world)
    gathered = np.zeros((world, outer, pad, inner), dtype=dtype)
    for r in range(world):
        lo, hi = shard_bounds(F, world, r)
        gathered[r, :, :hi - lo] = full[:, lo:hi]
        gathered[r, :, hi - lo:] = -7  # padding rows must never surface
    g = _lib.to_device(gathered)
    out = torch.empty((outer, F, inner), dtype=g.dtype, device=g.device)
    rc = _lib.load().pbbss_allgather_unpack(
        _lib.handle(g.device.index), _lib.ptr(g), g.element_size(), world, outer, F, inner,
        _lib.ptr(out), _lib.stream_ptr(g.device.index))
    assert rc == 0
    assert (_lib.to_host(out) == full).all()


def test_native_comm_world1_roundtrip():
    """pbbss_comm_unique_id -> pbbss_comm_create -> pbbss_allgather_masks on one rank: RCCL is
    resolved (dlopen), the communicator initialises, the gather is the identity; then
    pb_bss_amd.sharding.all_gather_bins takes the native path for CUDA tensors."""
    import torch
    import torch.distributed as dist
    from pb_bss_amd import _lib, sharding
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        sharding.init_native_comm()
        x = torch.randn(4, 9, 3, 11, dtype=torch.float64, device='cuda')
        for axis in (1, 0, -1):
            y = sharding.all_gather_bins(x, x.shape[axis], bin_axis=axis)
            assert y.shape == x.shape and bool((y == x).all())
        z = torch.randn(2, 5, 7, dtype=torch.complex128, device='cuda')
        assert bool((sharding.all_gather_bins(z, 5, bin_axis=1) == z).all())
        m = torch.randint(0, 3, (6, 3, 17), dtype=torch.int64, device='cuda')
        assert bool((sharding.all_gather_bins(m, 6, bin_axis=0) == m).all())
        f = torch.randn(2, 9, 4, dtype=torch.float32, device='cuda')
        assert bool((sharding.all_gather_bins(f, 9, bin_axis=1) == f).all())
        # a second communicator on the same handle is refused
        uid = ctypes.create_string_buffer(128)
        assert _lib.load().pbbss_comm_unique_id(uid) == 0
        assert _lib.load().pbbss_comm_create(_lib.handle(0), uid.raw, 1, 0) != 0
    finally:
        sharding.destroy_native_comm()
        if created:
            dist.destroy_process_group()


def test_allgather_without_comm_is_an_error():
    import torch
    from pb_bss_amd import _lib
    x = torch.zeros(1, 2, 3, dtype=torch.float64, device='cuda')
    out = torch.empty_like(x)
    rc = _lib.load().pbbss_allgather_masks(_lib.handle(0), _lib.ptr(x), 8, 1, 2, 3, _lib.ptr(out),
                                           _lib.stream_ptr(0))
    assert rc != 0


# ---------------------------------------------------------------------------------------------
# The sharded paths of round 3 at world size 1 (the one GPU the suite has): every collective is
# really enqueued on RCCL -- the mask all-gather, the all-reduce of the joint models' spectral
# M-step sums inside pbbss_joint_fit (pbbss_mix_opts.sharded), the all-reduce of the Souden
# reference-channel sums -- and must be the identity on the result.
@pytest.fixture
def world1_nccl():
    import torch
    import torch.distributed as dist
    from pb_bss_amd import sharding
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    sharding.init_native_comm()
    try:
        yield
    finally:
        sharding.destroy_native_comm()
        if created:
            dist.destroy_process_group()


def test_comm_info_reports_what_rccl_sees(world1_nccl):
    from pb_bss_amd import _lib
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    assert _lib.load().pbbss_comm_info(_lib.handle(0), ctypes.byref(w), ctypes.byref(r)) == 0
    assert (w.value, r.value) == (1, 0)


@pytest.mark.parametrize('kind,kw', [
    ('gaussian', dict()),
    ('gaussian', dict(weight_constant_axis=(-3, -1))),
    ('gaussian', dict(weight_constant_axis=(-3,), covariance_type='diagonal')),
    ('vmf', dict(weight_constant_axis=(-3, -1), max_concentration=80.)),
])
def test_sharded_joint_fit_world1_equals_oracle(world1_nccl, kind, kw):
    """fit_predict_sharded_joint: bins sliced, pbbss_joint_fit with opts.sharded = 1 (two-sweep
    first iteration, ncclAllReduce between partial sums and finalize, reduced class weights),
    masks gathered -- against the oracle's unsharded joint fit."""
    from pb_bss_amd.distribution import GCACGMMTrainer, VMFCACGMMTrainer
    from pb_bss_amd.distribution import _joint
    from pb_bss_amd import sharding
    from oracle import embed as oe, synth
    F, T, D, K, E = 12, 200, 4, 3, 16
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=11)
    sal = np.random.default_rng(1).uniform(0.2, 1.0, size=(F, T))
    trainer = GCACGMMTrainer() if kind == 'gaussian' else VMFCACGMMTrainer()
    # world size 1 would skip the in-library collective: force the sharded code path
    real = _joint.sharded_bins
    _joint.sharded_bins = lambda enabled=True: real(True)
    try:
        masks = sharding.fit_predict_sharded_joint(trainer, Y, e, init, iterations=6,
                                                   saliency=sal, **kw)
    finally:
        _joint.sharded_bins = real
    ref = oe.joint_fit(kind, Y.astype(np.complex128), e.astype(np.float64), init, 6,
                       saliency=sal, **kw)
    want = oe.joint_model_predict(ref, Y.astype(np.complex128), e.astype(np.float64))
    from pb_bss_amd import _lib
    got = _lib.to_host(masks)
    assert got.shape == (F, K, T)
    assert np.abs(got - want).max() < 1e-6


def test_sharded_joint_without_communicator_or_full_covariance_is_refused():
    from pb_bss_amd.distribution import GCACGMMTrainer, _joint
    from pb_bss_amd import _lib
    from oracle import synth
    Y, e, init = synth.make_joint(4, 80, 3, 2, 8, seed=1)
    with _joint.sharded_bins(True):
        with pytest.raises(_lib.PbbssError):  # no pbbss_comm_create on this handle
            GCACGMMTrainer().fit(Y, e, initialization=init, iterations=2)


def test_fit_predict_sharded_watson_and_souden_world1(world1_nccl):
    """BASELINE configs[3] under the sharded entry points: CWMMTrainer through
    fit_predict_sharded, then MVDR-Souden with the reference-channel sums all-reduced."""
    from pb_bss_amd.distribution import CWMMTrainer
    from pb_bss_amd import _lib, extraction as ex, pipeline, sharding
    from oracle import beamformer as ob, cwmm as ow, synth
    F, T, D, K = 17, 300, 6, 3
    Y, init = synth.make_stft(F, T, D, K, seed=4)
    Y128 = Y.astype(np.complex128)
    masks = sharding.fit_predict_sharded(_lib.to_device(Y), _lib.to_device(init), 10,
                                         trainer=CWMMTrainer())
    ref = ow.cwmm_predict(ow.cwmm_fit(Y128, init, iterations=10), Y128)
    assert np.abs(_lib.to_host(masks) - ref).max() < 1e-7
    psd = ex.get_power_spectral_density_matrix(_lib.to_device(Y).transpose(-1, -2).contiguous(), masks)
    psd_ref = ob.psd(Y128.transpose(0, 2, 1), ref)
    w, ch = ex.get_mvdr_vector_souden(psd[:, 0], psd[:, 1] + psd[:, 2], return_ref_channel=True,
                                      shard_group=True)
    w_ref, ch_ref = ob.mvdr_souden(psd_ref[:, 0], psd_ref[:, 1] + psd_ref[:, 2],
                                   return_ref_channel=True)
    assert ch == ch_ref
    assert np.abs(_lib.to_host(w) - w_ref).max() < 1e-6 * np.abs(w_ref).max()
    # the batched stage of pipeline.separate(beamformer='mvdr_souden'): (K, U, F, D, D)
    tgt = psd.movedim(1, 0).unsqueeze(1).contiguous()
    noi = (psd.sum(dim=1).unsqueeze(0).unsqueeze(0) - tgt).contiguous()
    wb = _lib.to_host(pipeline.device_ops.mvdr_souden(tgt, noi, True))
    wl = _lib.to_host(pipeline.device_ops.mvdr_souden(tgt, noi, None))
    assert np.abs(wb - wl).max() == 0.0
    for k in range(K):
        w_k = ob.mvdr_souden(psd_ref[:, k], psd_ref.sum(1) - psd_ref[:, k])
        assert np.abs(wb[k, 0] - w_k).max() < 1e-6 * np.abs(w_k).max()


def test_fit_predict_sharded_inline_aligner_world1(world1_nccl):
    """inline_permutation_aligner under bin sharding (sharding.sharded_inline_aligner): one mask
    all-gather per EM iteration through RCCL, the full mapping solved on the device, the local
    columns applied -- at world size 1 the result must equal the unsharded stepwise fit."""
    from pb_bss_amd import sharding
    from pb_bss_amd.distribution import CACGMMTrainer
    from pb_bss_amd.permutation_alignment import DHTVPermutationAlignment
    from oracle import synth
    F, T, D, K = 257, 90, 4, 2
    Y, init = synth.make_stft(F, T, D, K, seed=21)
    rng = np.random.default_rng(2)
    for f in range(F):  # a permuted initialisation gives the aligner something to do
        init[f] = init[f, rng.permutation(K)]
    aligner = DHTVPermutationAlignment.from_stft_size(512)
    kw = dict(iterations=4, weight_constant_axis=(-3,), inline_permutation_aligner=aligner)
    want = CACGMMTrainer().fit_predict(Y, initialization=init, **kw)
    got = sharding.fit_predict_sharded(Y, init, **kw)
    assert got.shape == (F, K, T)
    assert np.abs(got.cpu().numpy() - want).max() < 1e-12
    with pytest.raises(AssertionError):  # needs frequency-constant weights, as the reference
        sharding.fit_predict_sharded(Y, init, iterations=2, inline_permutation_aligner=aligner)
