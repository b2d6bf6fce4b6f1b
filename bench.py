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
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerfmeshes_amd import hip_ops, synthetic as S  # noqa: E402

from benchlib.common import (FAR, FP32_MFMA_PEAK_TFLOPS, H, MLP_KW, NEAR, NUM_COARSE, NUM_FINE, PARITY_RAYS, W, _free_port,  # noqa: E402
                             cpu_baseline)
from benchlib.guard import PORT_OVER_REFERENCE_TIME, _Emergency, _annotate_ports, _guarded, _inject  # noqa: E402,F401  (tests import them from here)
from benchlib.line import render as render_line, write_full  # noqa: E402
from benchlib.mesh import mesh_probe  # noqa: E402
from benchlib.scenes import b3_probe, buff_probe, eval_probe, tiny_probe  # noqa: E402
from benchlib.train import shapes_probe, train_probe  # noqa: E402


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
    for name in ("r06_pmc_mlp_kernel.json", "r05_pmc_mlp_kernel.json", "r04_pmc_mlp_kernel.json", "r03_pmc_mlp_kernel.json", "r02_pmc_mlp_kernel.json", "r01_pmc_mlp_kernel.json"):
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
    if world > 1:
        # what a SCALE record needs to be read on its own: the exchange step's size and count, and -- against the committed 1-GPU
        # line of this round, NOT measured in this run -- what fraction of N x the 1-GPU rate the job reached
        n1 = None
        for name in ("r06_bench_line.json", "r05_bench_line.json"):
            try:
                n1 = (float(json.load(open(os.path.join(ROOT, "profiles", name)))["value"]), "profiles/" + name)
                break
            except (OSError, ValueError, KeyError):
                continue
        out["scaling_detail"] = {"mode": args.mode, "collectives_per_step": 1,
                                 "gathered_bytes_per_step": int(gathered.numel() * 4) if gathered is not None else 0,
                                 "value_per_gpu": value / args.gpus,
                                 "efficiency_vs_n1_frac": value / (args.gpus * n1[0]) if n1 else None,
                                 "n1_value_source": n1[1] if n1 else None}

    full_path = os.path.join(ROOT, "bench_full.json")
    # the line is the compact form (<= 4 KB: the driver's record keeps it whole); every object as measured goes to bench_full.json
    emergency = _Emergency(json_fd, render=lambda o: render_line(o, "bench_full.json"), on_emit=lambda o: write_full(o, full_path))
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
                           ("train", args.no_train_probe, lambda: dict(train_probe(dev, views[0][1], views[0][0], cpu_legs=cpu_legs),
                                                                       shapes=shapes_probe(dev))),
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
