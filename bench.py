#!/usr/bin/env python3
"""Headline benchmark: rendered rays/sec of the NeRF ray-batch hot path on MI355X.

Workload (BASELINE.json configs[1]): nerf-synthetic-lego geometry -- 8x256 coarse + 8x256 fine
FlexibleNeRFModel, 64 coarse + 128 fine samples per ray, 800x800 views (640 000 rays), bounds [2, 6],
seeded synthetic weights / orbit poses (no dataset or checkpoint exists offline).  One "step" renders one
full view through the product path (nm_render_rays: coarse intervals -> coarse MLP -> composite ->
inverse-CDF resample -> fine MLP -> composite), ray directions already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU.  `python bench.py --gpus N` with no launcher environment re-executes itself
under `torch.distributed.run --nproc-per-node N` (and exits non-zero if the node has fewer than N GPUs); under
a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Default `--mode weak` (BASELINE config 3: the test set's views
dealt to the ranks): rank r renders view (step * N + r), followed by the one real exchange step of the path, an RCCL
all-gather of the rendered pixels (7.68 MB / rank / view).  `--mode strong`: every step is ONE view whose rays are
split into contiguous ranges over the ranks (`dist.render_view_sharded`), total work fixed.  At N > 1 the line also
carries the sharded `mesh` (config 4: axis-0 slabs of the density grid, all-gather, marching cubes; compute and
all-gather times separately) and `buff` (config 5: ray-sharded) objects with per-rank roofline fractions.
`--ranks-per-gpu k` (k > 1) is a FUNCTIONAL mode for boxes with fewer GPUs than ranks: k processes share each GPU over
the gloo backend (RCCL refuses duplicate devices); the line then says `"ranks_per_gpu": k` and its rates are not scaling
numbers.

Prints ONE JSON line on rank 0 (see the task contract): metric/value, roofline of the dominant kernel
(the fused MLP; fp32 MFMA peak 157.3 TFLOP/s), the CPU baseline (the oracle = torch-CPU port of the reference
path, timed on this host's cores on a bounded ray sample), PSNR parity against it, and -- at N = 1 -- secondary
objects for BASELINE configs 4 (`mesh`: 480^3 density grid + marching cubes) and 5 (`buff`: 504x378 rays x 192
samples through the voxel-tree sampler) and one training iteration (`train`).
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import hip_ops, synthetic as S  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0           # same guide, "HBM3E peak BW" (spec)
H = W = 800
NUM_COARSE, NUM_FINE = 64, 128
NEAR, FAR = 2.0, 6.0
MLP_KW = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
PARITY_RAYS = 32768


def _pick_threads(fn, ncpu):
    """torch's default of one thread per core is far from optimal for these problem sizes on a many-core host:
    try a few thread counts on one small call each and keep the fastest."""
    best = (float("inf"), ncpu)
    for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(threads)
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, threads)
    torch.set_num_threads(best[1])
    return best[1]


def cpu_baseline(weights, rays_o, rays_d, budget_s=16.0, chunk=2048):
    """Reference path on the host cores: the oracle (a torch-CPU restatement that is bit-identical to the
    reference's NeRFModel.forward) on chunks of 2048 rays (cfg.nerf.validation.chunksize).  The rate is taken over
    the first `budget_s` seconds; the remaining rays (up to rays_d.shape[0]) are rendered untimed for the parity
    check.  Returns (rays/s, timed rays, seconds, threads, reference rgb of ALL rays)."""
    from oracle import nerf_oracle as O   # cpu_baseline leg only
    ncpu = os.cpu_count() or 1
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
    o, d = rays_o.cpu(), rays_d.cpu()
    with torch.no_grad():
        threads = _pick_threads(lambda: O.render(weights, weights, spec, spec, rs, o, d[:512], NEAR, FAR), ncpu)
        done, outs, t0 = 0, [], time.perf_counter()
        timed = None
        while done < d.shape[0]:
            _, f = O.render(weights, weights, spec, spec, rs, o, d[done:done + chunk], NEAR, FAR)
            outs.append(f["rgb_map"])
            done += min(chunk, d.shape[0] - done)
            if timed is None and time.perf_counter() - t0 >= budget_s:
                timed = (done, time.perf_counter() - t0)
        if timed is None:
            timed = (done, time.perf_counter() - t0)
    return timed[0] / timed[1], timed[0], timed[1], threads, torch.cat(outs, 0)


def _events(n):
    return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]


def _timed(fn, reps):
    """(min ms, mean ms, last result) of `fn` with HIP events on torch's current stream (the stream every
    hip_ops wrapper launches on), after one warm-up call."""
    fn()
    torch.cuda.synchronize()
    ev, out = _events(reps), None
    for a, b in ev:
        a.record()
        out = fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    return min(ms), sum(ms) / len(ms), out


def train_probe(dev, dirs, origin, rays=2048, iters=10, cpu_rays=256, cpu_legs=True):
    """Secondary figure: one optimizer iteration of the same 8x256 coarse+fine model on a
    2048-ray batch -- forward in train mode (perturb + noise), MSE(coarse)+MSE(fine), HIP backward, Adam.  Roofline: the
    iteration's algorithmic fp32 matrix work -- forward, delta propagation (hidden columns of the transposed layers) and
    weight gradients, 524 288 samples x (1.187 + 1.114 + 1.187) MFLOP -- over the WHOLE iteration's wall time against the
    fp32 MFMA peak (so everything that is not a matrix kernel counts against it).  CPU leg: the same iteration through torch
    autograd over the oracle on a bounded ray batch."""
    from nerfmeshes_amd import models
    from nerfmeshes_amd.nerf import CfgNode
    torch.manual_seed(0)
    model = models.NeRFModel(CfgNode(S.hparams(train_perturb=True, train_noise_std=0.2))).to(dev)
    with torch.no_grad():
        for net in (model.model_coarse, model.model_fine):
            net.fc_alpha.weight.mul_(30.0)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    pick = torch.randperm(dirs.shape[0], generator=torch.Generator().manual_seed(1))[:rays].to(dev)
    batch = (origin.reshape(1, 3), dirs[pick].contiguous(), torch.tensor([2.0, 6.0]))
    target = torch.rand(rays, 3, device=dev)

    def iteration():
        opt.zero_grad(set_to_none=True)
        c, f = model(batch)
        loss = torch.nn.functional.mse_loss(c.rgb_map, target) + torch.nn.functional.mse_loss(f.rgb_map, target)
        loss.backward()
        opt.step()

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    kw = MLP_KW
    Hh, dx, dd = kw["hidden_size"], 6 * kw["num_encoding_fn_xyz"] + 3, 6 * kw["num_encoding_fn_dir"] + 3
    fwd = model.model_fine.hip().flops_per_sample()
    nskip = sum(1 for i in range(kw["num_layers"] - 1) if i % kw["skip_step"] == 0 and i > 0 and i != kw["num_layers"] - 1)
    delta = fwd - 2 * (dx * Hh * (1 + nskip) + dd * (Hh // 2) + Hh + 3 * (Hh // 2))      # no encoding columns, heads on the VALU
    samples = rays * (NUM_COARSE + NUM_COARSE + NUM_FINE)
    flops = samples * (fwd + delta + fwd)
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"value": rays / ms * 1e3, "unit": "rays/s", "ms_per_iteration": ms, "rays_per_iteration": rays,
           "workload": "training step: 8x256 coarse+fine, 64+128 samples, perturb + noise, Adam (forward + HIP backward + step)",
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "samples_per_iteration": samples,
                        "algorithmic_flops_per_sample": {"forward": fwd, "delta": delta, "weight_gradients": fwd},
                        "floor_ms_at_peak": flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                        "note": "whole-iteration wall time (taping forward, delta kernel, dW kernels, encodings, compositing, Adam) "
                                "against the fp32 MFMA peak"}}
    # ---- where the iteration's time goes (HIP events around the stages of train_ops, a separate pass of `iters` iterations):
    # the three matrix stages each against the fp32 MFMA peak on their own algorithmic FLOP, everything else as milliseconds
    from nerfmeshes_amd import train_ops
    train_ops.profile_stages(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    ms_prof = (time.perf_counter() - t0) / iters * 1e3
    stages = {k: v / iters for k, v in train_ops.profile_stages(False).items()}
    work = {"taping_forward": samples * fwd, "delta": samples * delta, "weight_gradients": samples * fwd}
    kernels = {}
    for name, ms_stage in sorted(stages.items(), key=lambda kv: -kv[1]):
        kernels[name] = {"ms": ms_stage}
        if name in work:
            kernels[name].update(tflops=work[name] / (ms_stage * 1e-3) / 1e12, frac=work[name] / (ms_stage * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS)
    kernels["rest"] = {"ms": ms_prof - sum(stages.values()),
                       "what": "sample_pdf, stratified jitter, random draws, the two MSE losses, autograd bookkeeping, parameter re-pack, Adam"}
    out["kernels"] = kernels
    out["kernels_note"] = (f"per-iteration averages over a separate pass of {iters} iterations with HIP events around the stages "
                           f"({ms_prof:.2f} ms per iteration in that pass); weight_gradients includes the order-fixed reductions and the "
                           "64-wide encoding products, head_gradients the fc_alpha / fc_rgb rows")
    if not cpu_legs:
        return out
    # ---- the same iteration through torch autograd over the CPU oracle, bounded
    from oracle import nerf_oracle as O
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(training=True)
    oc, dc, tgt = origin.reshape(1, 3).cpu(), batch[1][:cpu_rays].cpu(), target[:cpu_rays].cpu()
    wc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.model_coarse.named_parameters()}
    wf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.model_fine.named_parameters()}

    def cpu_iter(n=cpu_rays):
        t_c = O.perturb_intervals(O.coarse_intervals(NEAR, FAR, NUM_COARSE, n), torch.rand(n, NUM_COARSE))
        loss, tt = 0.0, t_c
        for w_, first in ((wc, True), (wf, False)):
            pts = O.ray_points(tt, dc[:n], oc).reshape(-1, 3)
            dirs_ = dc[:n, None, :].expand(-1, tt.shape[1], -1).reshape(-1, 3)
            rad = O.mlp_forward(w_, spec, pts, dirs_, keep_graph=True).reshape(n, -1, 4)
            b = O.composite(rad, tt, dc[:n], rs, noise=0.2 * torch.randn(n, tt.shape[1]))
            loss = loss + torch.nn.functional.mse_loss(b["rgb_map"], tgt[:n])
            if first:
                tt = O.sample_pdf_intervals(t_c, b["weights"].detach(), NUM_FINE, u=torch.rand(n, NUM_FINE))
        loss.backward()

    threads = _pick_threads(lambda: cpu_iter(64), os.cpu_count() or 1)      # the thread count is chosen on a quarter-size batch
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        cpu_iter()
    dt = (time.perf_counter() - t0) / reps
    out["cpu_baseline"] = {"value": cpu_rays / dt, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
                           "sample": f"forward + loss.backward() of the oracle (torch autograd, fp32) on {cpu_rays} rays, {reps} iterations, "
                                     f"{dt:.2f} s each (no optimizer step)"}
    out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def _wall_max(fn, dev, use_dist):
    """Wall time of `fn` between two (barrier +) device synchronisations, max over ranks; returns (seconds, result)."""
    from nerfmeshes_amd import dist as nd
    import torch.distributed as dist
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        nd.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def _per_rank(value, dev, world, use_dist):
    from nerfmeshes_amd import dist as nd
    if not use_dist:
        return [float(value)]
    mine = torch.tensor([[float(value)]], dtype=torch.float64, device=dev)
    return [float(x) for x in nd.all_gather_rows(mine, [1] * world).reshape(-1)]


def _mc_traffic_from_profile(res):
    """HBM bytes of one marching-cubes call from the committed PMC profile (480^3 only; never measured inside this run)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_mc_traffic.json")
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


def coarse_flops_per_sample():
    kw = MLP_KW
    H_, L_, dx, dd = kw["hidden_size"], kw["num_layers"], 6 * kw["num_encoding_fn_xyz"] + 3, 6 * kw["num_encoding_fn_dir"] + 3
    nskip = sum(1 for i in range(L_ - 1) if i % kw["skip_step"] == 0 and i > 0 and i != L_ - 1)
    return 2 * (dx * H_ + (L_ - 1) * H_ * H_ + nskip * dx * H_ + H_ * H_ + H_ + (H_ + dd) * (H_ // 2) + 3 * (H_ // 2))


def buff_probe(dev, cpu_rays=2048, rank=0, world=1, use_dist=False, cpu_legs=True):
    """BASELINE config 5 geometry (/root/reference/config/buff-colmap-fern.yml:31-74): BuFFModel.query on a
    504x378 view (fern 4032x3024 / 8), 192 samples per ray placed by the voxel-tree sampler (12^3 voxels on
    [-0.6, 0.6]^3), single 8x256 network, bounds [0, 1.2], synthetic pose on radius 1.  At N > 1 the view's rays are
    split into contiguous ranges over the ranks and the pixels all-gathered (strong scaling of one view)."""
    from nerfmeshes_amd import dist as nd, models
    hp = S.hparams(model="BuFFModel", use_fine=False, num_coarse=192, num_fine=64, near=0.0, far=1.2, dataset_type="colmap")
    w = S.make_mlp_weights(9, density_gain=1500.0, density_bias=60.0, **MLP_KW)
    model = models.BuFFModel(hp)
    sd = model.state_dict()
    for k, v in w.items():
        sd["model." + k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    hh, ww = 378, 504
    o, d = hip_ops.ray_bundle(S.pose_spherical(30.0, -20.0, 1.0), hh, ww, 0.8 * ww, device=dev)
    bounds = torch.tensor([0.0, 1.2])
    chunk = 65536
    rays = hh * ww
    lo, hi = nd.split_range(rays, rank, world)
    counts = [b_ - a_ for a_, b_ in (nd.split_range(rays, r, world) for r in range(world))]

    def view():
        outs = []
        for s0 in range(lo, hi, chunk):
            outs.append(model.query((o[None], d[s0:min(s0 + chunk, hi)], bounds)).rgb_map)
        mine = torch.cat(outs, 0)
        return nd.all_gather_rows(mine, counts) if world > 1 else mine

    with torch.no_grad():
        hip_ops.mlp_profile_enable(True)
        view()
        torch.cuda.synchronize()
        hip_ops.mlp_profile_read()
        reps = 3
        wall, rgb = _wall_max(lambda: [view() for _ in range(reps)][-1], dev, use_dist)
        wall /= reps
        launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
        hip_ops.mlp_profile_enable(False)
        timings = {}
        for tie in ("stable", "reference"):
            model.tree.tie_order = tie
            t_min, t_avg, (z, idx, mask) = _timed(lambda: model.tree.batch_ray_voxel_intersect(o[None], d[:chunk], 0.0, 1.2, 192), 5)
            timings[tie] = {"ms_min": t_min, "ms_avg": t_avg}
        model.tree.tie_order = "auto"
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {
        "workload": "buff-colmap-fern geometry: BuFFModel.query, 504x378 rays x 192 tree-placed samples, 8x256 network, "
                    + (f"rays split over {world} ranks + all-gather of the pixels" if world > 1 else "1 GPU"),
        "value": rays / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "rays_per_view": rays, "rays_per_rank": counts, "chunk_rays": chunk,
        "scaling": "strong" if world > 1 else None,
        "algorithmic_flops_per_ray": 192 * model.model.hip().flops_per_sample(),
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                     "frac_per_rank": _per_rank(achieved / FP32_MFMA_PEAK_TFLOPS, dev, world, use_dist),
                     "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / (wall * reps)},
        "nm_buff_intersect": {"rays": chunk, "voxels": int(model.tree.voxels.shape[0]), "samples": 192,
                              "rays_hitting_tree": float(mask.float().mean()),
                              "tie_order_stable": timings["stable"], "tie_order_reference": timings["reference"],
                              "note": "stable: every id is the voxel its sample lies in (eval default); reference: the reference's "
                                      "own ids (its three unstable sorts replayed), the default while training"},
    }
    if rank != 0 or not cpu_legs:
        return out
    # ---- CPU leg: the oracle's BuFF chain on a bounded strided ray sample + parity on those rays
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=192, num_fine=0)
    pick = torch.arange(0, rays, max(1, rays // cpu_rays), device=dev)[:cpu_rays]
    dd = d[pick].contiguous()
    vox = model.tree.voxels.detach().cpu()

    def cpu_render(dirs):
        zz, _, mm = O.buff_intersect(vox, o[None].cpu(), dirs, 0.0, 1.2, 192)
        uni = O.coarse_intervals(0.0, 1.2, 192, dirs.shape[0]).contiguous()
        zz = torch.where(mm[:, None], zz, uni)
        pts = O.ray_points(zz, dirs, o[None].cpu()).reshape(-1, 3)
        rad = O.mlp_forward(w, spec, pts, dirs[:, None, :].expand(-1, 192, -1).reshape(-1, 3)).reshape(dirs.shape[0], 192, 4)
        return O.composite(rad, zz, dirs, rs)["rgb_map"]

    with torch.no_grad():
        dc = dd.cpu()
        threads = _pick_threads(lambda: cpu_render(dc[:256]), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = torch.cat([cpu_render(dc[s:s + 1024]) for s in range(0, dc.shape[0], 1024)], 0)
        dt = time.perf_counter() - t0
        got = model.query((o[None], dd, bounds)).rgb_map.cpu()
    out["cpu_baseline"] = {"value": dc.shape[0] / dt, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(),
                           "kind": "port", "sample": f"{dc.shape[0]} strided rays of the view, chunks of 1024, {dt:.1f} s"}
    out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    out["parity"] = parity.psnr_parity(got, ref, chunk=1024)
    return out


def eval_probe(dev, weights, views=20, render_chunk=65536, cpu_views=4, cpu_size=92, cpu_legs=True):
    """BASELINE config 3 at N = 1 (`eval_nerf.py` over a test set, /root/reference/src/eval_nerf.py:50-105): `views` orbit
    views of 800x800 through the eval_nerf mirror (`eval_views`: per-view loss = sum of per-2048-ray-chunk MSEs divided by
    the FLOAT batch count 312.5, dataset loss = mean over views, PSNR of that), every view scored against a seeded noisy
    photograph of itself (~34 dB, the regime a trained NeRF is scored in).  Parity leg: `cpu_views` small views rendered by
    the oracle on the host, scored by the oracle's bookkeeping, against the same views through the mirror."""
    import contextlib, io
    from nerfmeshes_amd import eval_nerf as E, models
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
    gen = torch.Generator(device=dev)

    def photograph(view_nr, rgb):       # the view's own render + seeded noise, clamped: PSNR ~ 34 dB by construction
        gen.manual_seed(1000 + view_nr)
        return (rgb + 0.02 * torch.randn(rgb.shape, generator=gen, device=dev)).clamp_(0.0, 1.0)

    def run(n):
        vs = [(pose, H, W, S.LEGO_FOCAL_800, photograph) for pose in S.orbit_poses(n)]
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            return E.eval_views(model, vs, cfg, dev, render_chunk=render_chunk)

    run(1)
    torch.cuda.synchronize()
    hip_ops.mlp_profile_enable(True)
    hip_ops.mlp_profile_read()
    t0 = time.perf_counter()
    losses, total, psnr, _ = run(views)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    # what the UNMODIFIED script's loop shape costs: every call is one cfg.nerf.validation.chunksize = 2048-ray chunk
    # (eval_nerf.py:62-65), i.e. 313 calls of ~8 launches per view instead of 10
    vs1 = [(S.orbit_poses(views)[0], H, W, S.LEGO_FOCAL_800, photograph)]
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        E.eval_views(model, vs1, cfg, dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        small_losses, _, _, _ = E.eval_views(model, vs1, cfg, dev)
        torch.cuda.synchronize()
        wall_2048 = time.perf_counter() - t1
    out = {"workload": f"config 3 at N = 1: {views} orbit views of 800x800 through the eval_nerf mirror (eval_views), 8x256 coarse+fine, 64+128, "
                       f"rays generated in the kernels, rendered in calls of {render_chunk} rays, loss bookkeeping per 2048 rays / float batch_count 312.5",
           "value": views * H * W / wall, "unit": "rays/s", "views": views, "ms_per_view": wall / views * 1e3,
           "dataset_loss_mse": float(total), "dataset_psnr_db": float(psnr),
           "per_view_psnr_db_min_max": [float(min(-10.0 * torch.log10(l) for l in losses)), float(max(-10.0 * torch.log10(l) for l in losses))],
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                        "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / wall},
           "at_reference_chunksize": {"chunk_rays": int(cfg.nerf.validation.chunksize), "value": H * W / wall_2048, "unit": "rays/s",
                                      "ms_per_view": wall_2048 * 1e3, "same_loss_as_large_calls": bool(float(small_losses[0]) == float(losses[0])),
                                      "note": "one view rendered in the reference's own 2048-ray calls (313 per view): what the unmodified "
                                              "eval_nerf.py loop gets without raising nerf.validation.chunksize"}}
    if not cpu_legs:
        return out
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
    focal = S.LEGO_FOCAL_800 * cpu_size / 800.0
    small, ref_losses = [], []
    t0 = time.perf_counter()
    with torch.no_grad():
        for i, pose in enumerate(S.orbit_poses(views)[:: max(1, views // cpu_views)][:cpu_views]):
            o, d = O.get_ray_bundle(cpu_size, cpu_size, focal, torch.from_numpy(pose))
            d = d.reshape(-1, 3)
            ref = torch.cat([O.render(weights, weights, spec, spec, rs, o[None], d[s0:s0 + 2048], NEAR, FAR)[1]["rgb_map"]
                             for s0 in range(0, d.shape[0], 2048)])
            tgt = parity.noisy_targets(ref, seed=parity.TARGET_SEED + i)
            ref_losses.append(O.view_loss(ref, tgt, 2048))
            small.append((pose, cpu_size, cpu_size, focal, tgt))
    dt = time.perf_counter() - t0
    ref_total = O.dataset_loss(ref_losses)
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        got_losses, got_total, got_psnr, _ = E.eval_views(model, small, cfg, dev)
    rays = cpu_views * cpu_size * cpu_size
    out["cpu_baseline"] = {"value": rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
                           "sample": f"{cpu_views} views of {cpu_size}x{cpu_size} ({rays} rays; batch_count {cpu_size * cpu_size / 2048}), chunks of 2048, {dt:.1f} s"}
    out["parity"] = {"views": cpu_views, "rays_per_view": cpu_size * cpu_size, "float_batch_count": cpu_size * cpu_size / 2048,
                     "dataset_psnr_ref_db": float(O.mse2psnr(ref_total)), "dataset_psnr_hip_db": float(got_psnr),
                     "abs_dpsnr_db": abs(float(O.mse2psnr(ref_total)) - float(got_psnr)),
                     "per_view_abs_dpsnr_db": [abs(float(O.mse2psnr(a)) - float(O.mse2psnr(b.cpu()))) for a, b in zip(ref_losses, got_losses)],
                     "targets": "oracle render + N(0,0.02) PCG64, per view; HIP and oracle scored against the same targets by their own bookkeeping"}
    return out


def b3_probe(dev, weights, views, near, far, u_c, u_f, chunk, ref_idx, ref_rgb):
    """Opt-in precision mode "bf16x3" (every fp32 product emulated by six bf16 MFMA products of three-way operand
    splits, fp32 accumulation) on the headline workload: one 800x800 view, and its own PSNR parity against the SAME CPU
    reference render the fp32 path is scored on.  fp32 stays the default and the headline dtype."""
    from oracle import parity
    b3 = hip_ops.HipMLP(weights, MLP_KW, dev, precision="bf16x3")
    o, d = views[0]

    def view():
        for s in range(0, H * W, chunk):
            hip_ops.render_rays(b3, b3, o, d[s:s + chunk], near, far, u_c, u_f)

    hip_ops.mlp_profile_enable(True)
    view()
    torch.cuda.synchronize()
    hip_ops.mlp_profile_read()
    t0 = time.perf_counter()
    view()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    out = {"workload": "the headline view through the opt-in bf16x3 kernels (fp32-emulating: 3-way bf16 split of both "
                       "operands, 6 bf16 MFMA products, fp32 accumulation)",
           "value": H * W / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "dtype": "bf16x3",
           "algorithmic_tflops": kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0,
           "note": "algorithmic FLOP of the fp32 network / kernel time; the bf16 matrix pipe executes 6x as many"}
    if ref_rgb is not None:
        _, fb = hip_ops.render_rays(b3, b3, o, d[ref_idx].contiguous(), near, far, u_c, u_f)
        out["parity"] = parity.psnr_parity(fb["rgb_map"].cpu(), ref_rgb, chunk=2048)
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def tiny_probe(dev, cpu_legs=True):
    """BASELINE config 1 (`config/tiny.yaml` sizes: 4-layer x 64 MLP, 32 coarse samples, no fine network, ONE 400x400 view;
    the reference runs it on the CPU as plumbing): the same product path on the GPU, its MLP kernel against the fp32 MFMA
    roof (useful FLOP only -- the 64-wide layers pad their encodings), parity against the oracle on a ray sample, and the
    oracle timed on the host."""
    kw = dict(num_layers=4, hidden_size=64, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    w = S.make_mlp_weights(11, density_gain=30.0, density_bias=0.3, **kw)
    net = hip_ops.HipMLP(w, kw, dev)
    hh = ww = 400
    o, d = hip_ops.ray_bundle(S.orbit_poses(4)[1], hh, ww, S.LEGO_FOCAL_800 / 2, device=dev)
    near, far = torch.tensor([NEAR], device=dev), torch.tensor([FAR], device=dev)
    u_c = torch.linspace(0.0, 1.0, 32).to(dev)

    def view():
        return hip_ops.render_rays(net, None, o[None], d, near, far, u_c, None)[0]["rgb_map"]

    hip_ops.mlp_profile_enable(True)
    view()
    torch.cuda.synchronize()
    hip_ops.mlp_profile_read()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        rgb = view()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {"workload": "config 1 (tiny): 4x64 MLP, 32 coarse samples, no fine network, one 400x400 view, 1 GPU",
           "value": hh * ww / wall, "unit": "rays/s", "ms_per_view": wall * 1e3, "rays_per_view": hh * ww,
           "algorithmic_flops_per_ray": 32 * net.flops_per_sample(),
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "launches": launches,
                        "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / (wall * reps),
                        "note": "one 160 000-ray launch of 5.1 M samples lasts ~2 ms: the share of wall not in the MLP kernel is the per-ray kernels and launch latency"}}
    if not cpu_legs:
        return out
    from oracle import nerf_oracle as O, parity
    spec, rs = O.MLPSpec(**kw), O.RenderSpec(num_coarse=32, num_fine=0)
    idx = torch.arange(0, hh * ww, 10, device=dev)                       # 16 000 rays
    dc = d[idx].cpu()
    with torch.no_grad():
        threads = _pick_threads(lambda: O.render(w, None, spec, None, rs, o[None].cpu(), dc[:2048], NEAR, FAR), os.cpu_count() or 1)
        t0 = time.perf_counter()
        ref = torch.cat([O.render(w, None, spec, None, rs, o[None].cpu(), dc[s0:s0 + 2048], NEAR, FAR)[0]["rgb_map"]
                         for s0 in range(0, dc.shape[0], 2048)])
        dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": dc.shape[0] / dt, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
                           "sample": f"{dc.shape[0]} rays of the view (stride 10), chunks of 2048, {dt:.2f} s"}
    out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    out["parity"] = parity.psnr_parity(rgb[idx].cpu(), ref, chunk=2048)
    try:
        out["train"] = tiny_train_probe(dev)
    except Exception as e:      # a figure of a figure: never at the expense of the rest of the object
        out["train"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def tiny_train_probe(dev, rays=8192, iters=40):
    """Config 1's training iteration (4x64, 32 coarse samples, no fine network; perturb + noise, MSE, backward through the HIP
    kernels, Adam) -- about forty launches of a few tens of microseconds: launched eagerly, and replayed from one captured hipGraph
    (train_ops.GraphedStep; the same kernels: tests/test_gpu_train.py::test_training_iteration_replays_from_a_hipgraph)."""
    from nerfmeshes_amd import models, train_ops
    from nerfmeshes_amd.nerf import CfgNode
    hp = S.hparams(train_perturb=True, train_noise_std=0.2, hidden_size=64, num_layers=4, skip_step=2, num_encoding_fn_xyz=6,
                   num_encoding_fn_dir=4, num_coarse=32, num_fine=0, use_fine=False)
    g = torch.Generator().manual_seed(1)
    dirs = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1).to(dev)
    batch = (torch.tensor([[0.0, 0.0, 4.0]], device=dev), dirs, torch.tensor([NEAR, FAR]))
    target = torch.rand(rays, 3, generator=g).to(dev)

    def build(**adam):
        torch.manual_seed(0)
        model = models.NeRFModel(CfgNode(hp)).to(dev)
        model.train()
        opt = train_ops.make_optimizer("Adam", model.parameters(), 5e-4, **adam)

        def iteration():
            opt.zero_grad(set_to_none=True)
            out = model(batch)
            c = out[0] if isinstance(out, tuple) else out
            torch.nn.functional.mse_loss(c.rgb_map, target).backward()
            opt.step()
        return iteration

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    eager = timed(build())
    replay = timed(train_ops.GraphedStep(build(capturable=True)))
    return {"workload": f"config 1 training iteration: 4x64, {rays} rays x 32 samples, perturb + noise, fused Adam",
            "ms_per_iteration_eager": eager, "ms_per_iteration_graph_replay": replay, "rays_per_s_graph_replay": rays / replay * 1e3,
            "note": "launch-bound: one captured hipGraph replaces ~40 launches per iteration (train_ops.GraphedStep)"}


class _Emergency:
    """The headline must reach stdout whatever happens to a secondary object.  At N > 1 a rank that
    dies inside a sharded object leaves the others inside a collective; the launcher then SIGTERMs them.  Rank 0 therefore
    arms, as soon as the headline is computed, a watcher THREAD on the signal wake-up pipe (a Python-level handler would
    not run while the main thread sits in a collective / device synchronisation; those calls release the GIL, so a thread
    does): on SIGTERM / SIGINT it writes the ONE JSON line -- headline + {"error": ...} for the object in flight -- and exits."""

    def __init__(self, json_fd):
        self.json_fd, self.out, self.stage, self.done = json_fd, None, "headline", False

    def arm(self, out):
        import signal
        import threading
        self.out = out
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        for sig in (signal.SIGTERM, signal.SIGINT):
            signal.signal(sig, lambda *a: None)          # the C-level handler writes the signal number to the pipe

        def watch():
            os.read(r, 1)
            self.emit(f"the job was terminated during '{self.stage}' (a rank left it; signal from the launcher)", code=3)

        threading.Thread(target=watch, daemon=True).start()

    def emit(self, error=None, code=None):
        if self.done or self.out is None:
            if code is not None:
                os._exit(code)
            return
        self.done = True
        if error is not None:
            self.out.setdefault("errors", []).append(error)
            if self.stage not in self.out:
                self.out[self.stage] = {"error": error}
        _annotate_ports(self.out)
        os.write(self.json_fd, (json.dumps(self.out, default=repr) + "\n").encode())
        if code is not None:
            os._exit(code)


PORT_OVER_REFERENCE_TIME = 1.15   # profiles/r04_port_vs_reference_cpu.json: the oracle (kind "port") takes 1.14 - 1.16 x the time of the
                                  # unmodified reference modules on the same host (bit-identical outputs; first-touch of its activations)


def _annotate_ports(node):
    """Every CPU leg of kind "port" says by how much the port understates the reference's own CPU rate, so that no
    GPU / CPU ratio on the line is read as more than an upper bound."""
    if isinstance(node, dict):
        if node.get("kind") == "port" and "port_over_reference_time" not in node:
            node["port_over_reference_time"] = PORT_OVER_REFERENCE_TIME
            node["port_note"] = ("the port runs 1.14-1.16x the unmodified reference's time on the same host (profiles/r04_port_vs_reference_cpu.json): "
                                 "ratios against this leg are upper bounds by that factor")
        for v in list(node.values()):
            _annotate_ports(v)
    elif isinstance(node, list):
        for v in node:
            _annotate_ports(v)


def _inject(name, rank):
    """Test hook (tests/test_gpu_dist.py): NM_BENCH_INJECT_FAILURE="mesh:1" raises inside that object on that rank,
    "buff:all" on every rank."""
    spec = os.environ.get("NM_BENCH_INJECT_FAILURE", "")
    for item in spec.split(","):
        obj, _, who = item.partition(":")
        if obj == name and who in ("all", str(rank)):
            raise RuntimeError(f"injected failure in '{name}' on rank {rank}")


def _guarded(name, fn, rank, world, emergency, healthy_wait_s=1800, failed_wait_s=45):
    """Run one secondary object so that its failure cannot take the line down or hang the job.  The object's own
    collectives run on RCCL; the VERDICT on the object travels through the rendezvous store (no collective a failed rank could
    mismatch): every rank posts "" or its error after leaving the object and waits for the others' posts.
      * all ranks fail at the same place (a bug, an out-of-memory at this size): all post promptly, all skip together,
        the line carries {"error": ...} for the object and the next object runs;
      * one rank fails while the others sit in a collective it never joins: its wait for their posts times out
        (`failed_wait_s`), it leaves the job, the launcher terminates the rest and rank 0's emergency writer emits the line."""
    import datetime
    emergency.stage = name
    err, res = None, None
    try:
        _inject(name, rank)
        res = fn()
    except Exception as e:  # noqa: BLE001 -- the headline line must not depend on a secondary figure
        err = repr(e)
    if world == 1:
        return {"error": err} if err else res
    from torch.distributed.distributed_c10d import _get_default_store
    store = _get_default_store()
    keys = [f"nm_bench/{name}/{r}" for r in range(world)]
    store.set(keys[rank], err or "")
    try:
        store.wait(keys, datetime.timedelta(seconds=failed_wait_s if err else healthy_wait_s))
    except Exception:  # noqa: BLE001 -- the others never left the object: they are inside a collective this rank abandoned
        msg = f"rank {rank} failed in '{name}' ({err}) while other ranks were inside a collective" if err else \
              f"rank {rank}: other ranks never left '{name}'"
        if rank == 0:
            emergency.emit(msg, code=3)
        os._exit(3)
    errs = {r: store.get(k).decode() for r, k in enumerate(keys)}
    failed = {r: e for r, e in errs.items() if e}
    if failed:
        return {"error": next(iter(failed.values())), "failed_ranks": sorted(failed)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=65536, help="rays per nm_render_rays call")
    ap.add_argument("--mode", choices=("weak", "strong"), default="weak",
                    help="weak: one whole view per rank and step; strong: one view per step, its rays split over the ranks")
    ap.add_argument("--ranks-per-gpu", type=int, default=1,
                    help="functional mode: this many processes share each GPU over gloo (not a scaling measurement)")
    ap.add_argument("--mesh-res", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-probe", action="store_true")
    ap.add_argument("--no-mesh-probe", action="store_true")
    ap.add_argument("--no-buff-probe", action="store_true")
    ap.add_argument("--no-b3-probe", action="store_true")
    ap.add_argument("--no-tiny-probe", action="store_true")
    ap.add_argument("--no-eval-probe", action="store_true")
    ap.add_argument("--eval-views", type=int, default=20)
    ap.add_argument("--headline-only", action="store_true", help="skip every secondary object and the CPU legs")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_train_probe = args.no_mesh_probe = args.no_buff_probe = args.no_b3_probe = True
        args.no_tiny_probe = args.no_eval_probe = True

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback exists for the hot path)")
    visible = torch.cuda.device_count()
    if args.gpus < 1 or args.ranks_per_gpu < 1:
        raise SystemExit("--gpus and --ranks-per-gpu must be >= 1")
    if args.gpus > visible:
        raise SystemExit(f"--gpus {args.gpus} but only {visible} GPU(s) are visible on this node")
    ranks = args.gpus * args.ranks_per_gpu
    if args.ranks_per_gpu > 1:
        os.environ["NERFMESHES_RANKS_PER_GPU"] = str(args.ranks_per_gpu)
    if "WORLD_SIZE" not in os.environ and ranks > 1:
        # no launcher: become one.  One process per rank, rendezvous on 127.0.0.1.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    # stdout carries the ONE JSON line and nothing else: whatever native libraries print there (RCCL's version banner
    # at communicator creation / teardown, which lands AFTER the line) is sent to stderr by pointing fd 1 at fd 2 and
    # keeping the real stdout for the final write
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from nerfmeshes_amd import dist as nd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != ranks:
        raise SystemExit(f"--gpus {args.gpus} x --ranks-per-gpu {args.ranks_per_gpu} but WORLD_SIZE={world}")
    dist = None
    # a launcher environment (WORLD_SIZE set, also WORLD_SIZE=1) or NM_BENCH_FORCE_DIST=1 exercises the collective path
    use_dist = world > 1 or "WORLD_SIZE" in os.environ or os.environ.get("NM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()) if world == 1 else "29500")
        rank, _, dev = nd.init_from_env()     # RCCL ("nccl"), one rank per GPU; gloo when ranks share a GPU
    else:
        rank, dev = 0, torch.device("cuda", 0)
        torch.cuda.set_device(0)

    weights = S.make_scene_weights(**MLP_KW)
    coarse = hip_ops.HipMLP(weights, MLP_KW, dev)
    fine = hip_ops.HipMLP(weights, MLP_KW, dev)
    u_c = torch.linspace(0.0, 1.0, NUM_COARSE).to(dev)
    u_f = torch.linspace(0.0, 1.0, NUM_FINE).to(dev)
    near, far = torch.tensor([NEAR], device=dev), torch.tensor([FAR], device=dev)

    strong = args.mode == "strong"
    total_steps = args.warmup + args.steps
    views_per_step = 1 if strong else world
    poses = S.orbit_poses(max(total_steps * views_per_step, 1))
    lo, hi = nd.split_range(H * W, rank, world) if strong else (0, H * W)
    counts = [b - a for a, b in (nd.split_range(H * W, r, world) for r in range(world))] if strong else [H * W] * world
    views = []   # ray directions resident in HBM before the timed region (strong: this rank's range of the step's view)
    for s in range(total_steps):
        o, d = hip_ops.ray_bundle(poses[s if strong else s * world + rank], H, W, S.LEGO_FOCAL_800, lo, hi - lo, device=dev)
        views.append((o[None].contiguous(), d))
    image = torch.empty(hi - lo, 3, device=dev)
    gathered = None

    def step(i):
        nonlocal gathered
        o, d = views[i]
        for s in range(0, hi - lo, args.chunk):
            _, fb = hip_ops.render_rays(coarse, fine, o, d[s:s + args.chunk], near, far, u_c, u_f)
            image[s:s + args.chunk] = fb["rgb_map"]
        if use_dist:
            gathered = nd.all_gather_rows(image, counts)     # weak: `world` views; strong: the one view, in ray order

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    hip_ops.mlp_profile_enable(True)
    hip_ops.mlp_profile_read()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms, kernel_flops = hip_ops.mlp_profile_read()
    hip_ops.mlp_profile_enable(False)
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    rccl = None
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        nd.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what the all-gather actually delivered: every rank's slot of the last step must hold THAT rank's pixels
        # (slot checksums are compared with the checksums the ranks computed locally), and per-rank roofline fractions
        mine = torch.stack([image.double().sum(), torch.tensor(achieved, device=dev, dtype=torch.float64)])[None]
        allv = nd.all_gather_rows(mine.contiguous(), [1] * world)
        starts = [sum(counts[:r]) for r in range(world)]
        slots = torch.stack([gathered[starts[r]:starts[r] + counts[r]].double().sum() for r in range(world)])
        rccl = {"backend": dist.get_backend(), "ranks_in_all_gather": int(dist.get_world_size()),
                "gathered_bytes_per_step": int(gathered.numel() * 4),
                "slots_match_rank_checksums": bool(torch.allclose(slots, allv[:, 0], rtol=1e-9, atol=0)),   # fp64 sums, two reduction shapes
                "roofline_frac_per_rank": [float(x) / FP32_MFMA_PEAK_TFLOPS for x in allv[:, 1]]}

    rays_total = args.steps * H * W * views_per_step
    value = rays_total / elapsed
    flops_per_ray = (NUM_COARSE + NUM_COARSE + NUM_FINE) * coarse.flops_per_sample()

    traffic, traffic_source = None, None
    for name in ("r05_pmc_mlp_kernel.json", "r04_pmc_mlp_kernel.json", "r03_pmc_mlp_kernel.json", "r02_pmc_mlp_kernel.json", "r01_pmc_mlp_kernel.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/{name}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command "
                                  "(committed profile, NOT measured inside this run)")
                break
            except Exception:
                traffic = None

    if world == 1:
        parallelism = "single GPU"
    elif strong:
        parallelism = f"each view's rays split over {world} ranks (contiguous ranges), all-gather of pixels"
    else:
        parallelism = f"views dealt to {world} ranks, all-gather of pixels"
    out = {
        "metric": "rendered rays/sec (64+128 samples, 8x256 MLP), lego scene geometry",
        "value": value, "unit": "rays/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "nerf-synthetic lego: 8x256 coarse+fine MLP, 64 coarse + 128 fine samples, "
                               + ("one 800x800 view per step, rays split over the ranks" if strong else "800x800 view per step per rank")
                               + ", bounds [2,6], seeded weights/orbit poses",
                   "rays_per_step_per_rank": hi - lo, "chunk_rays": args.chunk, "parallelism": parallelism},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "nm::mlp_kernel3<256,10,4,8,8,1>", "launches": launches,
                     "avg_launch_ms": kernel_ms / max(launches, 1),
                     "algorithmic_flops_per_ray": flops_per_ray,
                     "mlp_kernel_share_of_wall": kernel_ms * 1e-3 / elapsed},
    }
    if args.ranks_per_gpu > 1:
        out["ranks_per_gpu"] = args.ranks_per_gpu
        out["note"] = (f"FUNCTIONAL run: {world} ranks share {args.gpus} GPU(s) over gloo (host-staged collectives); it exercises the "
                       "N-rank code paths on the real kernels and is not a scaling measurement")
    if rccl is not None:
        out["rccl"] = rccl

    emergency = _Emergency(json_fd)
    if rank == 0:
        emergency.arm(out)        # from here on the line is written even if the job is torn down around rank 0
    solo = rank == 0 and world == 1
    ref_idx = ref_rgb = None
    if solo and not args.no_cpu_baseline:
        from oracle import parity
        o, d = views[0]
        idx = torch.arange(0, H * W, (H * W) // PARITY_RAYS, device=dev)[:PARITY_RAYS]   # strided sample of the view
        rps, n_timed, dt, threads, ref_rgb = cpu_baseline(weights, o, d[idx])
        ref_idx = idx
        _, fb = hip_ops.render_rays(coarse, fine, o, d[idx].contiguous(), near, far, u_c, u_f)
        out["cpu_baseline"] = {"value": rps, "unit": "rays/s", "cores": threads, "host_cores": os.cpu_count(),
                               "kind": "port",
                               "sample": f"{n_timed} rays of view 0 (stride {(H * W) // PARITY_RAYS}), chunks of 2048, {dt:.1f} s "
                                         f"({threads} torch threads = fastest of 8/16/32/64/{os.cpu_count()} on this host)",
                               "speedup": value / rps}
        out["parity"] = parity.psnr_parity(fb["rgb_map"].cpu(), ref_rgb, chunk=2048)
        # which rays exceed 1e-4, and why (oracle/parity.py::explain_outliers): the declared ill-conditioned classes only
        err = (fb["rgb_map"].cpu() - ref_rgb).abs().max(-1).values
        bad = torch.nonzero(err > 1e-4).reshape(-1)
        if bad.numel():
            from oracle import nerf_oracle as O
            spec, rs = O.MLPSpec(**MLP_KW), O.RenderSpec(num_coarse=NUM_COARSE, num_fine=NUM_FINE)
            db = d[idx][bad.to(dev)].contiguous()
            rc, rf = O.render(weights, weights, spec, spec, rs, o.cpu(), db.cpu(), NEAR, FAR)
            t_c = hip_ops.coarse_intervals(u_c, near, far, db.shape[0])
            t_f = hip_ops.sample_pdf(t_c, hip_ops.composite(coarse.eval_rays(o, db, t_c), t_c, db)["weights"], u_f)
            t_r = rf["t"].to(dev).contiguous()
            on_ref = hip_ops.composite(fine.eval_rays(o, db, t_r), t_r, db)["rgb_map"]
            why = parity.explain_outliers(err[bad], rc, rf, t_f, on_ref)
            why.pop("unexplained_rays")
            out["parity"]["rays_over_1e-4_explained"] = why
    cpu_legs = world == 1 and not args.no_cpu_baseline
    shard = dict(rank=rank, world=world, use_dist=use_dist and world > 1, cpu_legs=cpu_legs)
    # objects every rank takes part in (sharded at N > 1) ...
    for name, skip, fn in (("mesh", args.no_mesh_probe, lambda: mesh_probe(dev, weights, fine, res=args.mesh_res, **shard)),
                           ("buff", args.no_buff_probe, lambda: buff_probe(dev, **shard))):
        if skip:
            continue
        res = _guarded(name, fn, rank, world, emergency)
        if rank == 0:
            out[name] = res
    # ... and single-GPU objects
    for name, skip, fn in (("eval", args.no_eval_probe, lambda: eval_probe(dev, weights, views=args.eval_views, cpu_legs=cpu_legs)),
                           ("tiny", args.no_tiny_probe, lambda: tiny_probe(dev, cpu_legs)),
                           ("train", args.no_train_probe, lambda: train_probe(dev, views[0][1], views[0][0], cpu_legs=cpu_legs)),
                           ("bf16x3", args.no_b3_probe,
                            lambda: b3_probe(dev, weights, views, near, far, u_c, u_f, args.chunk, ref_idx, ref_rgb))):
        if solo and not skip:
            out[name] = _guarded(name, fn, 0, 1, emergency)
    if use_dist:
        dist.barrier()
    if rank == 0:
        sys.stdout.flush()
        emergency.stage = "done"
        emergency.emit()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
