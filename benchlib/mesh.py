"""bench.py's `mesh` object (BASELINE config 4): density grid, marching cubes, per-vertex appearance, OBJ."""
import json
import os
import time

import torch

from nerfmeshes_amd import hip_ops, synthetic as S

from .common import (FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, MLP_KW, NUM_COARSE, NUM_FINE, ROOT, _per_rank, _pick_threads, _timed, _wall_max,
                     coarse_flops_per_sample)


def _mc_traffic_from_profile(res):
    """HBM bytes of one marching-cubes call from the committed PMC profile (480^3 only; never measured inside this run)."""
    path = os.path.join(ROOT, "profiles", "r03_mc_traffic.json")
    if res != 480 or not os.path.exists(path):
        return {"traffic": None}
    try:
        t = json.load(open(path))
        return {"traffic": t["total_bytes_fetch_x2_everywhere"],
                "traffic_source": "profiles/r03_mc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE passes over "
                                  "tests/tools/bench_mesh.py, all 8 kernels of a call (committed profile, NOT measured inside this run)"}
    except (OSError, ValueError, KeyError):
        return {"traffic": None}


def mesh_probe(dev, weights, fine, res=480, limit=1.2, iso_request=32.0, cpu_points=262144, rank=0, world=1,
               use_dist=False, cpu_legs=True):
    """BASELINE config 4 (`mesh_nerf.py --res 480 --limit 1.2 --iso-level 32`, /root/reference/src/mesh_nerf.py:27-92):
    the density-grid query (fused MLP, density-only trunk) against the fp32 MFMA roof -- rank r evaluates its slab of
    axis-0 planes (`dist.slab_range`), one all-gather assembles the grid on every rank -- then marching cubes against the
    HBM roof on its algorithmic bytes (4 B / voxel), the mesh compared bitwise IN THIS RUN with the C oracle on the same
    grid (rank 0), and at N = 1 the CPU legs (oracle MLP on a bounded point sample; oracle marching cubes)."""
    import numpy as np
    from nerfmeshes_amd import dist as nd
    from nerfmeshes_amd.mesh_nerf import extract_iso_level
    ax = torch.linspace(-limit, limit, res).to(dev)
    plane = res * res
    lo, hi = nd.slab_range(res, rank, world)
    counts = [(b - a) * plane for a, b in (nd.slab_range(res, r, world) for r in range(world))]
    slab = torch.empty((hi - lo) * plane, dtype=torch.float32, device=dev)
    query = lambda: fine.grid_query(ax, ax, ax, first=lo * plane, count=(hi - lo) * plane, density_only=True, out=slab)  # noqa: E731
    g_own, _, _ = _timed(query, 1)                           # this rank's kernel time (HIP events)
    g_wall, _ = _wall_max(query, dev, use_dist)              # slowest rank, wall
    if world > 1:
        nd.all_gather_rows(slab, counts)                     # warm the communicator / staging buffers
        a_wall, grid = _wall_max(lambda: nd.all_gather_rows(slab, counts), dev, use_dist)
    else:
        a_wall, grid = 0.0, slab
    density = grid.view(res, res, res)
    flops_own = (hi - lo) * plane * fine.flops_per_sample(density_only=True)
    frac_own = flops_own / (g_own * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS

    class _A:   # the script's adaptive iso level (mesh_nerf.py:56-65)
        iso_level = iso_request
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        iso = float(extract_iso_level(density, _A))
    m_min, m_avg, (v, f, n, val) = _timed(lambda: hip_ops.marching_cubes(density, iso), 5)
    vol_bytes = res ** 3 * 4
    total_flops = res ** 3 * fine.flops_per_sample(density_only=True)
    out = {
        "workload": f"mesh_nerf --res {res} --limit {limit} --iso-level {iso_request}: density grid + marching cubes, "
                    + (f"{world} ranks" if world > 1 else "1 GPU"),
        "grid_query": {"points": res ** 3, "ms": g_wall * 1e3, "planes_per_rank": [c // plane for c in counts],
                       "algorithmic_flops_per_point": fine.flops_per_sample(density_only=True),
                       "roofline": {"bound": "mfma", "achieved": total_flops / g_wall / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS * world,
                                    "unit": "TFLOP/s", "frac": total_flops / g_wall / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world),
                                    "frac_per_rank": _per_rank(frac_own, dev, world, use_dist),
                                    "note": "whole-job: all ranks' points / slowest rank's wall time, peak x ranks; per rank: own slab / own kernel time"}},
        "marching_cubes": {"iso": iso, "vertices": int(v.shape[0]), "faces": int(f.shape[0]), "ms_min": m_min, "ms_avg": m_avg,
                           "algorithmic_bytes": vol_bytes,
                           "roofline": {"bound": "hbm", "achieved": vol_bytes / (m_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": vol_bytes / (m_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,   # averages, as every other frac
                                        **_mc_traffic_from_profile(res)},
                           "note": "whole nm_mc_count + nm_mc_emit call on the full grid incl. workspace allocation and the host sync"},
    }
    if world > 1:
        # The two exchange strategies of the sharded script, each end to end (grid query + statistics + marching cubes +
        # collectives; max over ranks):
        #   "grid"       axis-0 slabs of the grid all-gathered, marching cubes on the whole grid on every rank;
        #   "triangles"  (mesh_nerf's default) every rank meshes its own cube layers (+ 2-3 recomputed ghost planes), only the
        #                vertices / faces / normals / values are all-gathered.
        from nerfmeshes_amd import mesh_nerf

        class _Model:
            @staticmethod
            def get_model():
                class _N:
                    hip = staticmethod(lambda precision=None: fine)
                return _N

        def run(gather):
            args = mesh_nerf.build_parser().parse_args(["--res", str(res), "--limit", str(limit), "--iso-level", str(iso_request), "--gather", gather])
            with contextlib.redirect_stdout(io.StringIO()):
                return mesh_nerf.extract_geometry(_Model, dev, args)

        strategies = {}
        for gather in ("grid", "triangles"):
            run(gather)
            wall, (gv, gf, gn, _) = _wall_max(lambda: run(gather), dev, use_dist)
            same = bool(torch.equal(gf, f) and torch.equal(gn, n) and gv.shape == v.shape)
            strategies[gather] = {"ms_end_to_end": wall * 1e3, "faces_and_normals_equal_single_grid_mesh": same}
        mesh_bytes = int(v.numel() * 4 + f.numel() * 4 + n.numel() * 4 + val.numel() * 4)
        out["sharded"] = {"strategies": strategies, "default": "triangles",
                          "all_gather_of_the_grid": {"ms": a_wall * 1e3, "bytes_total": vol_bytes,
                                                     "GBps_per_rank_received": (vol_bytes * (world - 1) / world) / a_wall / 1e9 if a_wall else None},
                          "all_gather_of_the_triangles": {"bytes_total": mesh_bytes},
                          "grid_gather_share_of_grid_strategy": a_wall * 1e3 / strategies["grid"]["ms_end_to_end"]}
    if rank != 0:
        return out
    # ---- CPU side: the checker (every N) and the baselines (N = 1 only)
    from oracle import mc_oracle, nerf_oracle as O
    vol = density.cpu().numpy()
    t0 = time.perf_counter()
    rv, rf, rn, rval = mc_oracle.marching_cubes(vol, iso)
    dt = time.perf_counter() - t0
    # the level itself: the GPU replays numpy's fp32 reductions (nm_np_stats) -> must equal numpy's own on the host copy
    iso_numpy = float(min(max(iso_request, vol.min() + vol.std()), vol.max() - vol.std()))
    out["marching_cubes"]["iso_equals_numpy_fp32"] = bool(iso == iso_numpy)
    same = (np.array_equal(rf, f.cpu().numpy()) and rv.tobytes() == v.cpu().numpy().tobytes()
            and rn.tobytes() == n.cpu().numpy().tobytes() and rval.tobytes() == val.cpu().numpy().tobytes())
    out["marching_cubes"]["bitwise_identical_to_oracle"] = bool(same)
    if not cpu_legs:
        return out
    out["marching_cubes"]["cpu_baseline"] = {"value": dt, "unit": "s", "cores": 1, "kind": "port",
                                             "sample": f"the full {res}^3 grid (oracle/mc_lewiner.c)"}
    spec = O.MLPSpec(**MLP_KW)
    pts = O.grid_points(limit, res)[:: max(1, res ** 3 // cpu_points)][:cpu_points]
    with torch.no_grad():
        threads = _pick_threads(lambda: O.mlp_forward(weights, spec, pts[:8192], pts[:8192]), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = O.mlp_forward(weights, spec, pts, pts)
        dt = time.perf_counter() - t0
        # the reference's own loop shape: batches of --batch-size 1024 points (mesh_nerf.py:43-48,239), bounded sample
        small = pts[:65536]
        t0 = time.perf_counter()
        for s0 in range(0, small.shape[0], 1024):
            O.mlp_forward(weights, spec, small[s0:s0 + 1024], small[s0:s0 + 1024])
        dt1024 = time.perf_counter() - t0
    got = fine.sample_points(pts.to(dev), pts.to(dev)).cpu()
    out["grid_query"]["cpu_baseline"] = {"value": pts.shape[0] / dt, "unit": "points/s", "cores": threads,
                                         "host_cores": os.cpu_count(), "kind": "port",
                                         "sample": f"{pts.shape[0]} strided grid points, one batch, {dt:.2f} s",
                                         "at_reference_batch_1024": {"value": small.shape[0] / dt1024, "unit": "points/s",
                                                                     "sample": f"{small.shape[0]} points in batches of 1024 (mesh_nerf.py --batch-size default), {dt1024:.2f} s"}}
    out["grid_query"]["at_reference_batch_1024"] = reference_batch_probe(dev, weights, fine, res, limit)
    out["grid_query"]["speedup_vs_cpu"] = (res ** 3 / g_wall) / (pts.shape[0] / dt)
    scale = float(ref[:, 3].abs().max()) + 1.0
    out["grid_query"]["parity"] = {"max_abs_dsigma_over_scale": float((got[:, 3] - ref[:, 3]).abs().max()) / scale,
                                   "max_abs_drgb": float((got[:, :3] - ref[:, :3]).abs().max()), "points": int(pts.shape[0])}
    out["appearance"], out["end_to_end_s"] = appearance_probe(dev, weights, res, limit, iso_request)
    # ---- end-to-end topology against the CPU path (mesh_nerf.py:73-79) at a size the oracle's grid takes seconds for:
    # HIP grid -> GPU iso level -> nm_mc_* vs oracle grid -> numpy iso level -> C marching cubes
    from oracle import parity
    tres = 128
    tax = torch.linspace(-limit, limit, tres).to(dev)
    tgrid = fine.grid_query(tax, tax, tax, density_only=True).view(tres, tres, tres)
    with contextlib.redirect_stdout(io.StringIO()):
        tiso = float(extract_iso_level(tgrid, _A))
    tmesh = [t.cpu().numpy() for t in hip_ops.marching_cubes(tgrid, tiso)]
    with torch.no_grad():
        t0 = time.perf_counter()
        rgrid = O.extract_radiance(weights, spec, limit, tres)[..., 3]
        dt = time.perf_counter() - t0
    riso = float(O.iso_level(rgrid, iso_request))
    topo = parity.mesh_topology(tgrid.cpu().numpy(), rgrid, tiso, riso, tmesh, mc_oracle.marching_cubes(np.ascontiguousarray(rgrid), riso))
    topo["cpu_grid_s"] = dt
    out["parity"] = {"topology": topo,
                     "note": f"end to end at {tres}^3: the HIP density grid meshed by nm_mc_* vs the oracle's CPU grid meshed by the C oracle, "
                             "each at its own adaptive iso level; on an identical grid the two marching cubes agree bitwise (marching_cubes."
                             "bitwise_identical_to_oracle)"}
    return out


def reference_batch_probe(dev, weights, fine, res, limit, planes=16, batch=1024):
    """Route A of INTEGRATION.md at the script's OWN defaults: what the unmodified mesh_nerf.py does to the HIP path
    (/root/reference/src/mesh_nerf.py:37-48, `--batch-size` default 1024 at :239) -- the grid points built on the host, `batchify`
    moving 1024 of them to the device per call, `model.sample_points` through the module surface (FlexibleNeRFModel.forward ->
    hip(): one parameter re-pack + one fused-MLP launch per call), `.cpu()` per batch -- on a bounded run of `planes` whole axis-0
    planes of the res^3 grid, extrapolated to the 108 000 calls of the full grid.  Launch- and copy-bound by construction: the
    figure says what a maintainer gets WITHOUT touching the script; nm_mlp_grid_query above is what the mirror's route B gets."""
    from nerfmeshes_amd import models
    from nerfmeshes_amd.nerf.nerf_helpers import batchify
    model = models.NeRFModel(S.hparams())
    sd = model.state_dict()
    for k, v in weights.items():
        sd["model_coarse." + k] = torch.from_numpy(v)
        sd["model_fine." + k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    tiles = [torch.linspace(-limit, limit, res)] * 3
    first = (res // 2) * res * res                      # planes through the middle of the scene
    samples = torch.stack(torch.meshgrid(*tiles, indexing="ij"), -1).view(-1, 3).float()[first:first + planes * res * res]

    def run(points):
        got = []
        with torch.no_grad():
            for (x,) in batchify(points, batch_size=batch, device=dev, progress=False):
                got.append(model.sample_points(x, x).cpu())
        return torch.cat(got, 0)

    run(samples[:8 * batch])
    torch.cuda.synchronize()
    before = model.model_fine.refresh_count()
    t0 = time.perf_counter()
    rad = run(samples)
    dt = time.perf_counter() - t0
    calls = (samples.shape[0] + batch - 1) // batch
    one = fine.sample_points(samples.to(dev), samples.to(dev)).cpu()
    return {"value": samples.shape[0] / dt, "unit": "points/s", "calls": calls, "ms_per_call": dt / calls * 1e3,
            "full_grid_s": res ** 3 / (samples.shape[0] / dt), "full_grid_calls": (res ** 3 + batch - 1) // batch,
            "parameter_repacks_per_call": (model.model_fine.refresh_count() - before) / calls,
            "same_sigma_as_one_call": bool(torch.equal(rad, one)),
            "sample": f"{planes} axis-0 planes of the {res}^3 grid ({samples.shape[0]} points) in batches of {batch}: host points -> batchify "
                      "-> model.sample_points -> .cpu() per batch, as the unmodified script's loop"}


def appearance_probe(dev, weights, res, limit, iso_request, max_bound=1.0, cpu_rays=2048):
    """The rest of BASELINE config 4 (next row (f)-1, /root/reference/src/mesh_nerf.py:131-201 + export_obj,
    src/nerf/nerf_helpers.py:86-111): `export_marching_cubes` of the mirror, whole, on the README's command
    (`--res 480 --iso-level 32 --limit 1.2 --view-disparity-max-bound 1e0`) -- geometry, the per-vertex appearance re-query
    (a full coarse+fine ray per vertex from v + 0.01 n along -n, per-ray origins), the OBJ text -- timed stage by stage;
    then the `--no-view-dependence` branch (one network evaluation per vertex).  Roofline of the re-query: V rays x
    303.8 MFLOP against the fp32 MFMA peak over the stage's wall time.  CPU leg: the oracle on a bounded vertex sample,
    which is also the parity check."""
    import contextlib, io, tempfile
    from nerfmeshes_amd import mesh_nerf, models
    from nerfmeshes_amd.nerf import CfgNode
    from nerfmeshes_amd.models.model_helpers import nest_dict
    hp = S.hparams()
    model = models.NeRFModel(hp)
    sd = model.state_dict()
    for k, v in weights.items():
        sd["model_coarse." + k] = torch.from_numpy(v)
        sd["model_fine." + k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    cfg = CfgNode(nest_dict(hp, sep="."))
    tmp = tempfile.mkdtemp(prefix="nm_bench_mesh_")
    stages, kept = {}, {}

    def timed(name, fn):
        def wrapper(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res_ = fn(*a, **k)
            torch.cuda.synchronize()
            stages[name] = stages.get(name, 0.0) + time.perf_counter() - t0
            return res_
        return wrapper

    orig = mesh_nerf.export_obj, mesh_nerf.extract_geometry
    mesh_nerf.export_obj, mesh_nerf.extract_geometry = timed("obj_text_s", orig[0]), timed("geometry_s", orig[1])
    out = {}
    try:
        for branch, extra in (("view_dependent", []), ("no_view_dependence", ["--no-view-dependence"])):
            args = mesh_nerf.build_parser().parse_args(["--res", str(res), "--iso-level", str(iso_request), "--limit", str(limit),
                                                        "--view-disparity-max-bound", str(max_bound), "--save-dir", tmp] + extra)
            for _ in range(2):                 # the second run is the warm one
                stages.clear()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
                    vertices, triangles, normals, diffuse = mesh_nerf.export_marching_cubes(model, args, cfg, dev)
                torch.cuda.synchronize()
                total = time.perf_counter() - t0
            V = int(vertices.shape[0])
            requery = total - stages["geometry_s"] - stages["obj_text_s"]
            size = os.path.getsize(os.path.join(tmp, args.mesh_name))
            flops = V * (256 * coarse_flops_per_sample() if branch == "view_dependent" else coarse_flops_per_sample())
            out[branch] = {"vertices": V, "faces": int(triangles.shape[0]), "end_to_end_s": total, "geometry_s": stages["geometry_s"],
                           "requery_s": requery, "requery_rays_per_s": V / requery,
                           "roofline": {"bound": "mfma", "achieved": flops / requery / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": flops / requery / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                        "note": "V x algorithmic FLOP of the branch over the stage's WALL time (D2H of the colours and host glue included)"},
                           "obj_text_s": stages["obj_text_s"], "obj_bytes": size, "obj_MBps": size / stages["obj_text_s"] / 1e6}
            kept[branch] = (vertices, normals, torch.as_tensor(diffuse))
    finally:
        mesh_nerf.export_obj, mesh_nerf.extract_geometry = orig
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    # ---- CPU leg + parity on a bounded, strided vertex sample: the oracle's NeRFModel.query over per-ray origins
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
    vertices, normals, diffuse = kept["view_dependent"]
    pick = torch.arange(0, vertices.shape[0], max(1, vertices.shape[0] // cpu_rays))[:cpu_rays]
    tgt, dirs = vertices[pick.to(dev)].cpu(), -normals[pick.to(dev)].cpu()
    origins = tgt - 0.01 * dirs
    with torch.no_grad():
        threads = _pick_threads(lambda: O.render(weights, weights, spec, spec, rs, origins[:256], dirs[:256], 0.0, max_bound), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = O.render(weights, weights, spec, spec, rs, origins, dirs, 0.0, max_bound)[1]["rgb_map"]
        dt = time.perf_counter() - t0
        ref_pts = O.mlp_forward(weights, spec, tgt, dirs)[:, :3]
    out["view_dependent"]["cpu_baseline"] = {"value": pick.numel() / dt, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(),
                                             "kind": "port", "sample": f"{pick.numel()} strided vertices, per-ray origins, one call, {dt:.2f} s"}
    out["view_dependent"]["speedup_vs_cpu"] = out["view_dependent"]["requery_rays_per_s"] / (pick.numel() / dt)
    out["view_dependent"]["parity"] = parity.psnr_parity(diffuse[pick], ref, chunk=2048)
    out["no_view_dependence"]["parity"] = {"max_abs_drgb": float((kept["no_view_dependence"][2][pick] - ref_pts).abs().max()),
                                           "vertices": int(pick.numel())}
    out["workload"] = (f"mesh_nerf --res {res} --iso-level {iso_request} --limit {limit} --view-disparity-max-bound {max_bound}: "
                       "export_marching_cubes of the mirror, whole (geometry + per-vertex re-query + OBJ), warm run")
    return out, out["view_dependent"]["end_to_end_s"]
