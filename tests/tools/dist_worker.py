"""Launched by tests/test_gpu_dist.py under `python -m torch.distributed.run --nproc-per-node N` on the GPU box:
joins the process group through nerfmeshes_amd.dist.init_from_env -- RCCL (backend "nccl", one rank per GPU), or gloo
with `NERFMESHES_RANKS_PER_GPU=N` (N ranks sharing ONE GPU: the N-rank code paths on the real kernels where the box
has a single device) -- and pushes REAL kernel outputs through the sharding layer: rendered pixels
(render_view_sharded: contiguous ray ranges per rank), a density slab (density_grid_sharded: axis-0 planes per rank) ->
marching cubes, the sharded per-vertex appearance re-query of mesh_nerf, the ragged per-view loss gather of eval_nerf,
and the gradient all-reduce of a real training step.  Every rank checks that the N-rank result equals the
single-process one bit for bit.  Prints DIST_OK on rank 0."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import dist as nd, hip_ops, synthetic as S  # noqa: E402


def single_rank(fn):
    """Run `fn` as a process outside any group would (nd.world() -> (0, 1)): the 1-rank result, computed on this rank."""
    real = nd.world, nd.all_gather_rows
    nd.world, nd.all_gather_rows = (lambda: (0, 1)), (lambda local, counts: local)
    try:
        return fn()
    finally:
        nd.world, nd.all_gather_rows = real


def model_level_checks(rank, world, dev):
    """The three consumers of the sharding layer inside the package, each against its own single-rank result."""
    import argparse
    import contextlib
    import io
    import tempfile
    from nerfmeshes_amd import eval_nerf, mesh_nerf, models
    hp = S.hparams(chunksize=3000, train_perturb=True, train_noise_std=0.0)
    torch.manual_seed(0)                                     # identical replicas on every rank
    model = models.NeRFModel(hp)
    sd = model.state_dict()
    for prefix in ("model_coarse.", "model_fine."):
        for k, v in S.make_scene_weights().items():
            sd[prefix + k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    quiet = contextlib.redirect_stdout(io.StringIO())

    # (1) eval_nerf: 3 views on `world` ranks (ragged; with 2 ranks rank 1 holds one view), losses back in view order
    views = list(eval_nerf.synthetic_views(3, height=36, width=44, focal=60.0))
    with torch.no_grad(), quiet:
        l1, t1, _, _ = single_rank(lambda: eval_nerf.eval_views(model, views, model.cfg, dev))
        ln, tn, _, _ = eval_nerf.eval_views(model, views, model.cfg, dev)
    assert len(ln) == 3 and all(float(a) == float(b) for a, b in zip(l1, ln)) and float(t1) == float(tn), "eval losses differ"

    # (2) mesh_nerf: slab-sharded density grid -> marching cubes -> vertex-sharded appearance re-query -> OBJ on rank 0
    # both exchange strategies: per-slab marching cubes + all-gather of the triangles (default), and the all-gathered grid;
    # res 44: planes narrower than numpy's 8192-element chunks (statistics from the assembled grid), res 100: the
    # chunk-sum path; every variant must reproduce the 1-rank mesh bit for bit, vertex numbering included
    out = {}
    for tag, res, gather in (("n", "44", "triangles"), ("g", "44", "grid"), ("1", "44", "triangles"),
                             ("N", "100", "triangles"), ("I", "100", "triangles")):
        d = tempfile.mkdtemp(prefix=f"nm_dist_{rank}_{tag}_")
        args = mesh_nerf.build_parser().parse_args(["--res", res, "--save-dir", d, "--view-disparity-max-bound", "1.0",
                                                    "--iso-level", "32", "--batch-size", "4096", "--gather", gather])
        with torch.no_grad(), quiet:
            run = lambda: mesh_nerf.export_marching_cubes(model, args, model.cfg, dev)   # noqa: E731
            out[tag] = (single_rank(run) if tag in "1I" else run()) + (d,)
    for multi, single in (("n", "1"), ("g", "1"), ("N", "I")):
        for a, b in zip(out[multi][:3], out[single][:3]):
            assert a.shape == b.shape and torch.equal(a, b), f"sharded mesh ({multi}) differs from the 1-rank mesh"
        assert (out[multi][3] == out[single][3]).all() and out[multi][3].shape[0] == out[multi][0].shape[0], "sharded vertex colours differ"
        if rank == 0:
            assert open(os.path.join(out[multi][4], "mesh.obj"), "rb").read() == open(os.path.join(out[single][4], "mesh.obj"), "rb").read()
        else:
            assert not os.path.exists(os.path.join(out[multi][4], "mesh.obj")), "only rank 0 writes the OBJ"

    # (3) training: every rank draws its own rays; the all-reduced gradients are the mean of the ranks' own, and the
    # replicas stay identical after the optimizer step
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(100 + rank)
    o, d = hip_ops.ray_bundle(S.orbit_poses(4)[2], 64, 64, 90.0, device=dev)
    pick = torch.randperm(4096, generator=g)[:1024].to(dev)
    batch = {"ray_origins": o, "ray_directions": d[pick], "ray_targets": torch.rand(1024, 3, generator=g).to(dev),
             "ray_bounds": torch.tensor([2.0, 6.0])}
    model.training_step(batch, 0)["loss"].backward()
    params = [p for p in model.parameters() if p.requires_grad]
    own = torch.cat([p.grad.reshape(-1) for p in params])
    nd.all_reduce_gradients(model.parameters())
    got = torch.cat([p.grad.reshape(-1) for p in params])
    every = nd.all_gather_rows(own[None].contiguous(), [1] * world)
    mean = every.sum(0) / world
    # two addends commute exactly; with more ranks the reduction order is the backend's: the bound is relative to the largest
    # ADDEND of an entry (gradients of different ranks cancel, so the mean can be far smaller than what was summed)
    bound = 1e-6 * every.abs().max(0).values + 1e-12
    assert torch.equal(got, mean) if world <= 2 else bool(((got - mean).abs() <= bound).all()), \
        f"all-reduced gradients are not the mean of the ranks' gradients: worst {float(((got - mean).abs() / bound).max()):.2f} x the bound"
    assert world == 1 or not torch.equal(every[0], every[-1]), "ranks were meant to train on different rays"
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    copies = nd.all_gather_rows(flat[None].contiguous(), [1] * world)
    assert all(torch.equal(copies[0], copies[r]) for r in range(world)), "replicas diverged after the optimizer step"
    return int(out["N"][0].shape[0])


def slab_rehearsal(rank, world, dev, mlp):
    """Per-slab marching cubes the way an 8-GPU node will run it: the 480 axis-0 planes of config 4 at
    reduced n1 / n2, cube layers dealt to the ranks (`dist.marching_cubes_sharded`), ONLY the triangles gathered
    (all-gatherv: exact sizes).  The surface lives in part of the slabs, so the ranks' shards are ragged and some are empty;
    a second, 5-plane grid has fewer cube layers than an 8-rank group has ranks (ranks without any layer still enter every
    collective).  Both must equal the 1-rank mesh bit for bit, vertex numbering included.  Returns the ranks' face counts."""
    out = []
    for n0, n1, n2 in ((480, 24, 20), (5, 24, 20)):
        ax0 = torch.linspace(-1.2, 1.2, n0).to(dev)
        ax1, ax2 = torch.linspace(-0.30, 0.30, n1).to(dev), torch.linspace(-0.25, 0.25, n2).to(dev)
        plane = n1 * n2
        whole = mlp.grid_query(ax0, ax1, ax2, density_only=True).view(n0, n1, n2)
        iso = float(whole.float().mean() + 0.5 * whole.float().std())
        want = hip_ops.marching_cubes(whole, iso)
        query = lambda p_lo, p_hi: mlp.grid_query(ax0, ax1, ax2, first=p_lo * plane, count=(p_hi - p_lo) * plane, density_only=True)   # noqa: E731
        v, f, nrm, val, slab = nd.marching_cubes_sharded(query, n0, n1, n2, lambda slab, p_lo, own_lo, own_hi: iso)
        for name, a, b in zip(("vertices", "faces", "normals", "values"), (v, f, nrm, val), want):
            assert a.shape == b.shape and torch.equal(a, b), f"{n0} planes on {world} ranks: {name} differ from the 1-rank mesh"
        # this rank's own share of the faces: those whose first vertex it owns is not defined -- count by layer instead
        lo, hi, below, above, p_lo, p_hi = nd.slab_layers(n0, rank, world)
        mine = 0
        if hi > lo:
            piece = hip_ops.marching_cubes_slab(slab, iso, p_lo, below, above)
            mine = piece.faces
        counts = nd.all_gather_rows(torch.tensor([[mine]], dtype=torch.int64, device=dev), [1] * world).reshape(-1).tolist()
        assert sum(counts) == f.shape[0], (counts, f.shape)
        out.append(counts)
    return out


def main():
    rank, world, dev = nd.init_from_env()
    import torch.distributed as dist
    want = os.environ.get("NM_EXPECT_BACKEND", "nccl")
    assert dist.is_initialized() and dist.get_backend() == want, f"{want} process group expected, got {dist.get_backend()}"
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
    if world == 1:      # the ragged path's collective (a broadcast per non-empty shard) on a one-rank RCCL communicator as well
        assert torch.equal(nd.all_gather_v(full[:1001].contiguous(), [1001]), full[:1001])
    per = min(4096, hh * ww // world)
    even = nd.render_view_sharded(render, per * world)       # equal shards: single-collective fast path
    assert torch.equal(even, full[:per * world])

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

    shards = slab_rehearsal(rank, world, dev, mlp)
    if world > 1:
        assert len(set(shards[0])) > 1, f"the 480-plane rehearsal is meant to produce ragged triangle shards: {shards[0]}"
    if world >= 8:
        assert 0 in shards[0] and shards[1].count(0) >= world - 4, f"empty shards / ranks without a cube layer expected: {shards}"
    mesh_vertices = model_level_checks(rank, world, dev)

    # barrier + max-over-ranks reduction as bench.py uses them
    t = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
    nd.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == world - 1
    dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        print(f"DIST_OK world={world} backend={dist.get_backend()} device={dev} rays={hh * ww} planes={n} "
              f"vertices={int(v1[0].shape[0])} mesh_vertices={mesh_vertices} faces_per_rank_480={shards[0]} faces_per_rank_5={shards[1]}", flush=True)
    nd.shutdown()


if __name__ == "__main__":
    main()
