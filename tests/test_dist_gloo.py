"""CPU, world_size 2, gloo: the multi-GPU sharding layer (nerfmeshes_amd.dist) assembles exactly what a single
process produces -- rays / grid slabs are disjoint contiguous ranges, one all-gather at the end."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerfmeshes_amd import dist as nd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def _fake_pixels(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.float32)
    return torch.stack([idx, idx * 0.5, idx + 1000.0], -1)


def _render_job(rank, world):
    full = nd.render_view_sharded(_fake_pixels, 1001)          # ragged: 501 + 500 rays
    even = nd.render_view_sharded(_fake_pixels, 640)           # equal shards: single-collective fast path
    return full.numpy(), even.numpy()


def _grid_job(rank, world):
    n0, n1, n2 = 7, 3, 5

    def query(lo, hi):
        return torch.arange(lo * n1 * n2, hi * n1 * n2, dtype=torch.float32) * 2.0

    return nd.density_grid_sharded(query, n0, n1, n2).numpy()


def test_split_range_is_a_partition():
    for n in (0, 1, 7, 640000, 480):
        for w in (1, 2, 3, 8):
            parts = [nd.split_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    assert nd.world() == (0, 1)
    out = nd.render_view_sharded(_fake_pixels, 10)
    assert torch.equal(out, _fake_pixels(0, 10))


@pytest.mark.timeout(120)
def test_render_view_sharded_two_ranks_equals_single():
    res = _run(_render_job)
    for full, even in res:
        assert np.array_equal(full, _fake_pixels(0, 1001).numpy())
        assert np.array_equal(even, _fake_pixels(0, 640).numpy())


@pytest.mark.timeout(120)
def test_density_grid_sharded_two_ranks_equals_single():
    res = _run(_grid_job)
    ref = (torch.arange(7 * 3 * 5, dtype=torch.float32) * 2.0).view(7, 3, 5).numpy()
    for grid in res:
        assert np.array_equal(grid, ref)


def _grad_job(rank, world):
    torch.manual_seed(0)                                        # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    unused = torch.nn.Parameter(torch.zeros(4))                 # a parameter without a gradient on rank 0
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) * (rank + 1) / 10.0
    loss = net(x).pow(2).mean() + (unused.sum() if rank == 1 else 0.0)
    loss.backward()
    local = [p.grad.clone() for p in net.parameters()]
    nd.all_reduce_gradients(list(net.parameters()) + [unused], bucket_bytes=64)    # several buckets
    return [g.numpy() for g in local], [p.grad.numpy() for p in net.parameters()], unused.grad.numpy()


def test_gradient_all_reduce_two_ranks_is_the_mean():
    res = _run(_grad_job)
    for k in range(4):
        mean = 0.5 * (res[0][0][k] + res[1][0][k])
        for r in range(2):
            np.testing.assert_allclose(res[r][1][k], mean, rtol=1e-6, atol=1e-7)
    for r in range(2):
        np.testing.assert_allclose(res[r][2], np.full(4, 0.5, dtype=np.float32))
