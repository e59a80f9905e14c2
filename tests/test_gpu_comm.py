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
