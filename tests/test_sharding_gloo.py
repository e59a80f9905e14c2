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
