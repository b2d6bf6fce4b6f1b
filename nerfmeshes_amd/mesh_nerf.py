"""Mesh extraction with the reference's function names, arguments and CLI
(/root/reference/src/mesh_nerf.py:27-283): dense-grid density query -> adaptive iso level -> Lewiner
marching cubes -> per-vertex appearance -> OBJ, all on the MI355X.

    python -m nerfmeshes_amd.mesh_nerf --log-checkpoint <logdir>/<exp>/<run>/version_N --res 480 --iso-level 32

Differences in mechanism (not in results): the (res^3, 3) sample tensor is never materialised (grid points are
generated in the MLP kernel from the three axis arrays), the radiance stays in HBM (no per-batch D2H), and
marching cubes runs on the GPU grid directly.  `extract_radiance` still returns the same (n0,n1,n2,4) fp32
array for callers that want it; `extract_geometry` uses the density-only path (the colour branch of the MLP is
skipped; sigma is bit-identical to the full evaluation).

`--route script` (addition) runs the UNMODIFIED script's own call sequence over the same kernels instead -- host-built sample
points, `batchify` at `--batch-size`, `model.sample_points` + `.cpu()` per batch, numpy's iso level,
`skimage.measure.marching_cubes`, the per-vertex re-query in `--batch-size` calls -- i.e. INTEGRATION.md's route A without the
reference checkout: what a maintainer gets who changes nothing (tests/test_gpu_script_traces.py compares its call trace with the
one recorded from the reference's script; bench.py times its inner loop as `mesh.grid_query.at_reference_batch_1024`).
"""
import argparse
import os

import torch

from . import hip_ops, models
from .lightning_modules import PathParser
from .nerf.nerf_helpers import batchify, export_obj


def create_mesh(vertices, faces_idx):
    """mesh_nerf.py:14-24: the mesh centred and scaled into the unit sphere as a pytorch3d `Meshes` (only the chamfer
    branch of `validation_epoch_end` uses it; needs pytorch3d)."""
    from pytorch3d.structures import Meshes
    vertices = vertices - vertices.mean(0)
    return Meshes(verts=[vertices / max(vertices.abs().max(0)[0])], faces=[faces_idx])


def _nums(nums):
    assert isinstance(nums, (tuple, list, int)), "Nums arg should be either iterable or int."
    if isinstance(nums, int):
        return (nums,) * 3
    assert len(nums) == 3, "Nums arg should be of length 3, number of axes for 3D"
    return tuple(int(n) for n in nums)


def _axes(args, nums, device):
    # mesh_nerf.py:37: torch.linspace on the host, then meshgrid 'ij' with the last axis fastest
    return [torch.linspace(-args.limit, args.limit, n).to(device) for n in nums]


def _grid_query(model, args, device, nums, density_only, shard=None):
    """shard = (rank, world): evaluate only this rank's slab of axis-0 planes (see nerfmeshes_amd.dist)."""
    nums = _nums(nums)
    net = model.get_model().hip("f32")
    ax = _axes(args, nums, device)
    first, count = 0, nums[0] * nums[1] * nums[2]
    if shard is not None:
        from .dist import slab_range
        lo, hi = slab_range(nums[0], *shard)
        first, count = lo * nums[1] * nums[2], (hi - lo) * nums[1] * nums[2]
    return net.grid_query(ax[0], ax[1], ax[2], first=first, count=count, density_only=density_only), nums


def extract_radiance(model, args, device, nums):
    """(n0,n1,n2,4) numpy fp32 [rgb, raw sigma], as mesh_nerf.py:27-53."""
    if getattr(args, "route", "kernel") == "script":
        # the script's own loop (mesh_nerf.py:37-51): the (N,3) sample tensor on the host, one sample_points call and one D2H
        # copy per `--batch-size` points
        nums = _nums(nums)
        tiles = [torch.linspace(-args.limit, args.limit, n) for n in nums]
        samples = torch.stack(torch.meshgrid(*tiles, indexing="ij"), -1).view(-1, 3).float()
        parts = [model.sample_points(x, x).cpu() for (x,) in batchify(samples, batch_size=args.batch_size, device=device)]
        return torch.cat(parts, 0).view(*nums, 4).contiguous().numpy()
    out, nums = _grid_query(model, args, device, nums, density_only=False)
    return out.view(*nums, 4).cpu().numpy()


def extract_density(model, args, device, nums):
    """Density grid (n0,n1,n2) as a GPU tensor (radiance[..., 3] of the reference, mesh_nerf.py:73).
    Under torch.distributed (one process per GPU) every rank evaluates its slab of axis-0 planes and one
    all-gather (RCCL over xGMI) assembles the full grid on every rank; marching cubes then runs on the full
    grid, so the mesh is identical to the single-GPU one by construction."""
    from . import dist as nd
    rank, world = nd.world()
    if world == 1:
        out, nums = _grid_query(model, args, device, nums, density_only=True)
        return out.view(*nums)
    nums = _nums(nums)
    net = model.get_model().hip("f32")
    ax = _axes(args, nums, device)
    plane = nums[1] * nums[2]
    return nd.density_grid_sharded(
        lambda lo, hi: net.grid_query(ax[0], ax[1], ax[2], first=lo * plane, count=(hi - lo) * plane, density_only=True),
        *nums)


def extract_iso_level(density, args):
    """mesh_nerf.py:56-65.  `density` may be a numpy array (the reference's own code path) or a GPU tensor: then the
    statistics are numpy's fp32 reductions replayed on the device (nm_np_stats: 8192-element chunks, pairwise sums,
    fp32 mean / variance) -- the SAME iso value bit for bit, which is what "bit-identical topology for the same
    iso-level" needs whenever the clamp to [min + std, max - std] is active."""
    if isinstance(density, torch.Tensor):
        st = hip_ops.np_stats(density)
        lo, hi, std, mean = st["min"], st["max"], st["std"], st["mean"]
    else:
        lo, hi, std, mean = density.min(), density.max(), density.std(), density.mean()
    iso_value = min(max(args.iso_level, lo + std), hi - std)
    print(f"Min density {lo}, Max density: {hi}, Mean density {mean}")
    print(f"Querying based on iso level: {iso_value}")
    return iso_value


def extract_iso_level_sharded(slab, first_plane, own_lo, own_hi, nums, args):
    """`extract_iso_level` of a grid that lives in slabs on several ranks: numpy's fp32 statistics of the WHOLE grid from
    per-rank chunk sums (hip_ops.np_stats_sharded), so that every rank gets the single-GPU level bit for bit."""
    from . import dist as nd
    n_total = nums[0] * nums[1] * nums[2]
    plane = nums[1] * nums[2]
    dev = slab.device if slab is not None else torch.device("cuda", torch.cuda.current_device())
    if plane < 8192:
        # numpy's 8192-element chunks are wider than the one-plane halo: a grid this small is simply assembled
        own = slab.reshape(-1)[own_lo - first_plane * plane:own_hi - first_plane * plane] if slab is not None else torch.empty(0, device=dev)
        st = hip_ops.np_stats(nd.all_gather_ragged(own.contiguous()))
    else:
        x = slab if slab is not None else torch.zeros(1, dtype=torch.float32, device=dev)
        st = hip_ops.np_stats_sharded(x, first_plane * plane, n_total, own_lo, own_hi, nd.all_gather_ragged)
    iso_value = min(max(args.iso_level, st["min"] + st["std"]), st["max"] - st["std"])
    print(f"Min density {st['min']}, Max density: {st['max']}, Mean density {st['mean']}")
    print(f"Querying based on iso level: {iso_value}")
    return iso_value, st


def extract_geometry(model, device, args):
    """mesh_nerf.py:68-92 -> (vertices (V,3) f32, triangles (F,3) i32, normals (V,3) f32, density grid).
    Under torch.distributed the default is `--gather triangles`: every rank meshes its own slab of cube layers and only
    the triangles travel (dist.marching_cubes_sharded; the 4th value is then the rank's own slab of the grid);
    `--gather grid` assembles the density grid on every rank first (extract_density) and meshes it redundantly."""
    from . import dist as nd
    rank, world = nd.world()
    if getattr(args, "route", "kernel") == "script":
        # mesh_nerf.py:68-92 as written: the full radiance grid on the host, numpy's statistics, scikit-image's entry point
        # (the real package, or compat's nm_mc_* stand-in where it is not installed)
        import numpy as np
        try:
            from skimage import measure
            marching_cubes = measure.marching_cubes
        except ImportError:
            from .compat import marching_cubes
        density = extract_radiance(model, args, device, args.res)[..., 3]
        iso_value = extract_iso_level(density, args)
        vertices, triangles, normals, _ = [torch.from_numpy(np.ascontiguousarray(r)) for r in marching_cubes(density, iso_value)]
        vertices = args.limit * (vertices / (args.res / 2.0) - 1.0)
        return vertices, triangles, normals, density
    if world > 1 and getattr(args, "gather", "triangles") == "triangles":
        nums = _nums(args.res)
        net = model.get_model().hip("f32")
        ax = _axes(args, nums, device)
        plane = nums[1] * nums[2]
        stats = {}

        def iso_fn(slab, p_lo, own_lo, own_hi):
            iso, st = extract_iso_level_sharded(slab, p_lo, own_lo, own_hi, nums, args)
            stats.update(st)
            return iso

        vertices, triangles, normals, _, slab = nd.marching_cubes_sharded(
            lambda p_lo, p_hi: net.grid_query(ax[0], ax[1], ax[2], first=p_lo * plane, count=(p_hi - p_lo) * plane, density_only=True),
            *nums, iso_fn)
        if vertices.shape[0] == 0:              # skimage's two errors, in its order (see hip_ops.marching_cubes)
            iso = min(max(args.iso_level, stats["min"] + stats["std"]), stats["max"] - stats["std"])
            if iso < stats["min"] or iso > stats["max"]:
                raise ValueError("Surface level must be within volume data range.")
            raise RuntimeError("No surface found at the given iso value.")
        vertices = args.limit * (vertices / (args.res / 2.0) - 1.0)
        return vertices, triangles, normals, slab
    density = extract_density(model, args, device, args.res)
    iso_value = extract_iso_level(density, args)
    vertices, triangles, normals, _ = hip_ops.marching_cubes(density, iso_value)
    vertices = args.limit * (vertices / (args.res / 2.0) - 1.0)   # res/2, not (res-1)/2: as the reference
    return vertices, triangles, normals, density


def _assemble_grid_from_slabs(slab, nums, device):
    """The whole density grid from the slabs `--gather triangles` left on the ranks (only the mesh cache wants it): every
    rank contributes the planes it accounts for -- the lower planes of its own cube layers, the last non-empty rank also the
    top plane -- through ONE ragged all-gather; nothing is evaluated twice.  A rank without a cube layer (more ranks than
    layers) holds no slab (None) and contributes nothing, but still enters the collective."""
    from . import dist as nd
    rank, world = nd.world()
    n0, n1, n2 = nums
    lo, hi, below, above, p_lo, p_hi = nd.slab_layers(n0, rank, world)
    if slab is None or hi == lo:
        own = torch.empty(0, n1, n2, dtype=torch.float32, device=device)
    else:
        top = hi + (1 if hi == n0 - 1 else 0)
        own = slab.view(p_hi - p_lo, n1, n2)[lo - p_lo:top - p_lo]
    return nd.all_gather_ragged(own.contiguous()).view(n0, n1, n2)


def extract_geometry_with_super_sampling(model, device, args):
    raise NotImplementedError   # dead code in the reference as well (mesh_nerf.py:95-96)


def export_marching_cubes(model, args, cfg, device):
    """mesh_nerf.py:131-201."""
    if args.super_sampling >= 1:
        return extract_geometry_with_super_sampling(model, device, args)
    from . import dist as nd
    cache_path = os.path.join(args.save_dir, args.cache_name)
    cached = os.path.exists(cache_path)
    cache_new = args.use_cached_mesh and not cached
    if args.use_cached_mesh and cached:
        print("Loading cached mesh geometry...")
        vertices, triangles, normals, density = torch.load(cache_path, weights_only=False)
        vertices, triangles, normals = (torch.as_tensor(t).to(device) for t in (vertices, triangles, normals))
    else:
        print("Generating mesh geometry...")
        vertices, triangles, normals, density = extract_geometry(model, device, args)
        if cache_new or args.override_cache_mesh:
            if nd.world()[1] > 1 and getattr(args, "gather", "triangles") == "triangles":
                density = _assemble_grid_from_slabs(density, _nums(args.res), device)   # the cache holds the whole grid
            if nd.world()[0] == 0:
                torch.save((vertices.cpu(), triangles.cpu(), normals.cpu(),
                            density.cpu().numpy() if isinstance(density, torch.Tensor) else density), cache_path)
                print(f"Cached mesh geometry saved to {cache_path}")

    if getattr(args, "precision", "f32") != "f32":     # the geometry above is fp32 by contract; only the colours may use the mode
        model.set_precision(args.precision)
    # Appearance: one query per vertex.  Vertices are independent, so under torch.distributed every rank queries a
    # contiguous range of them and one ragged all-gather assembles the (V,3) colours; rank 0 writes the file.
    rank, world = nd.world()
    counts = [b - a for a, b in (nd.split_range(vertices.shape[0], r, world) for r in range(world))]
    lo, hi = nd.split_range(vertices.shape[0], rank, world)
    targets, directions = vertices[lo:hi], -normals[lo:hi]
    diffuse = []
    # --batch-size bounds the reference's per-call memory (default 1024); on a 288 GB device the per-call overhead
    # of a 1024-ray launch sequence dominates, so at least 65 536 vertices go into one call
    chunk = max(int(args.batch_size), 65536)
    script = getattr(args, "route", "kernel") == "script"
    if script:
        if world > 1:
            raise ValueError("--route script is the unmodified script's single-process call sequence: run it on one rank")
        chunk = int(args.batch_size)                       # the script's calls as they are (mesh_nerf.py:172-191), D2H per batch
    keep = (lambda t: t.cpu()) if script else (lambda t: t)
    if args.no_view_dependence:
        print("Diffuse map query directly  without specific-views...")
        for pos, dirs in batchify(targets, directions, batch_size=chunk, device=device, progress=False):
            diffuse.append(keep(model.sample_points(pos, dirs)[..., :3]))
    else:
        print("Diffuse map query with view dependence...")
        ray_bounds = torch.tensor([0.0, args.view_disparity_max_bound], dtype=directions.dtype)
        ray_origins = targets - args.view_disparity * directions
        for o, d in batchify(ray_origins, directions, batch_size=chunk, device=device, progress=False):
            diffuse.append(keep(model.query((o, d, ray_bounds)).rgb_map))
    diffuse = torch.cat(diffuse, dim=0) if diffuse else torch.empty(0, 3, dtype=torch.float32, device=device)
    diffuse = nd.all_gather_rows(diffuse.contiguous(), counts).cpu().numpy()
    if getattr(args, "precision", "f32") != "f32":
        model.set_precision("f32")
    if rank == 0:
        export_obj(vertices.cpu(), triangles.cpu(), diffuse, normals.cpu(), os.path.join(args.save_dir, args.mesh_name))
    return vertices, triangles, normals, diffuse


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--log-checkpoint", type=str, default=None)
    p.add_argument("--checkpoint", type=str, default="model_last.ckpt")
    p.add_argument("--save-dir", type=str, default=".")
    p.add_argument("--mesh-name", type=str, default="mesh.obj")
    p.add_argument("--iso-level", type=float, default=32)
    p.add_argument("--limit", type=float, default=1.2)
    p.add_argument("--res", type=int, default=128)
    p.add_argument("--super-sampling", type=int, default=0)
    p.add_argument("--batch-size", type=int, default=1024)
    p.add_argument("--no-view-dependence", action="store_true", default=False)
    p.add_argument("--view-disparity", type=float, default=1e-2)
    p.add_argument("--view-disparity-max-bound", type=float, default=4e0)
    p.add_argument("--use-cached-mesh", action="store_true", default=False)
    p.add_argument("--override-cache-mesh", action="store_true", default=False)
    p.add_argument("--cache-name", type=str, default="mesh_cache.pt")
    p.add_argument("--precision", choices=("f32", "bf16x3"), default="f32",
                   help="(addition) arithmetic of the per-vertex appearance re-query; the density grid -- hence the mesh topology "
                        "-- is always computed in fp32")
    p.add_argument("--route", choices=("kernel", "script"), default="kernel",
                   help="(addition) kernel: one density-grid launch, GPU statistics and marching cubes (default); script: the "
                        "unmodified script's own call sequence -- sample_points + D2H per --batch-size points, numpy iso level, "
                        "skimage.measure.marching_cubes, the re-query in --batch-size calls (single process)")
    p.add_argument("--gather", choices=("triangles", "grid"), default="triangles",
                   help="(addition, multi-GPU) what travels between the ranks: the emitted triangles of per-slab marching "
                        "cubes (default) or the density grid")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    path_parser = PathParser()
    cfg, _ = path_parser.parse(None, args.log_checkpoint, None, args.checkpoint)
    if not torch.cuda.is_available():
        raise SystemExit("mesh_nerf needs a MI355X: the HIP path has no CPU fallback")
    from . import dist as nd
    rank, world, device = nd.init_from_env()          # one process per GPU under torch.distributed.run
    print(f"Loading model from {path_parser.checkpoint_path}")
    model = getattr(models, cfg.experiment.model).load_from_checkpoint(path_parser.checkpoint_path)
    model = model.eval().to(device)
    try:
        with torch.no_grad():
            return export_marching_cubes(model, args, cfg, device)
    finally:
        nd.shutdown()


if __name__ == "__main__":
    main()
