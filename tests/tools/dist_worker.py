"""Launched by tests/test_gpu_dist.py under `python -m torch.distributed.run --nproc-per-node N` on the GPU box:
initialises RCCL (backend "nccl") through nerfmeshes_amd.dist.init_from_env and pushes REAL kernel outputs through
the sharding layer -- rendered pixels (render_view_sharded: contiguous ray ranges per rank), a density slab
(density_grid_sharded: axis-0 planes per rank), the per-view eval losses (ragged) and a gradient all-reduce -- and
checks on EVERY rank that the N-rank result equals the single-process one bit for bit.  Prints DIST_OK on rank 0."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import dist as nd, hip_ops, synthetic as S  # noqa: E402


def main():
    rank, world, dev = nd.init_from_env()
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == "nccl", "RCCL process group expected"
    kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    w = S.make_scene_weights(**kw)
    mlp = hip_ops.HipMLP(w, kw, dev)
    hh, ww = 123, 157                                       # 19 311 rays: ragged for every world size > 1
    o, d = hip_ops.ray_bundle(S.orbit_poses(4)[1], hh, ww, S.LEGO_FOCAL_800 * ww / 800, device=dev)
    near, far = torch.tensor([2.0], device=dev), torch.tensor([6.0], device=dev)
    uc, uf = torch.linspace(0, 1, 64).to(dev), torch.linspace(0, 1, 128).to(dev)

    def render(lo, hi):
        _, fb = hip_ops.render_rays(mlp, mlp, o[None], d[lo:hi].contiguous(), near, far, uc, uf)
        return fb["rgb_map"].clone()

    full = render(0, hh * ww)                                # what one process produces
    sharded = nd.render_view_sharded(render, hh * ww)        # ragged all-gather over RCCL
    assert sharded.shape == full.shape and torch.equal(sharded, full), "sharded pixels differ from the 1-rank render"
    even = nd.render_view_sharded(render, 4096 * world)      # equal shards: single-collective fast path
    assert torch.equal(even, full[:4096 * world])

    n = 37                                                   # 37 planes: ragged slabs
    ax = torch.linspace(-1.2, 1.2, n).to(dev)
    plane = n * n
    grid1 = mlp.grid_query(ax, ax, ax, density_only=True).view(n, n, n)
    gridn = nd.density_grid_sharded(
        lambda lo, hi: mlp.grid_query(ax, ax, ax, first=lo * plane, count=(hi - lo) * plane, density_only=True), n, n, n)
    assert torch.equal(gridn, grid1), "sharded density grid differs from the 1-rank grid"
    iso = float(grid1.median())
    v1 = hip_ops.marching_cubes(grid1, iso)
    vn = hip_ops.marching_cubes(gridn, iso)
    assert all(torch.equal(a, b) for a, b in zip(v1, vn)), "mesh differs"

    # gradient all-reduce: identical replicas, rank-dependent data -> the mean on every rank
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4).to(dev)
    x = torch.arange(32, dtype=torch.float32, device=dev).reshape(4, 8) * (rank + 1) / 10.0
    lin(x).pow(2).mean().backward()
    local = lin.weight.grad.clone()
    nd.all_reduce_gradients(lin.parameters())
    gathered = nd.all_gather_rows(local[None].contiguous(), [1] * world)
    assert torch.allclose(lin.weight.grad, gathered.mean(0), rtol=1e-6, atol=1e-7)

    # barrier + max-over-ranks reduction as bench.py uses them
    t = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == world - 1
    dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        print(f"DIST_OK world={world} backend={dist.get_backend()} rays={hh * ww} planes={n} "
              f"vertices={int(v1[0].shape[0])}", flush=True)
    nd.shutdown()


if __name__ == "__main__":
    main()
