"""bench.py's training objects: `train` (one optimizer iteration of the 8x256 pair) and config 1's 4x64 iteration."""
import os
import time

import torch

from nerfmeshes_amd import synthetic as S

from .common import FAR, FP32_MFMA_PEAK_TFLOPS, MLP_KW, NEAR, NUM_COARSE, NUM_FINE, _pick_threads


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
    from nerfmeshes_amd import train_ops
    opt = train_ops.make_optimizer("Adam", model.parameters(), 5e-4)     # what BaseModel.configure_optimizers builds (fused on the GPU)
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
    # round 6: layer1's gradient is taken by linearity (no activation follows layer1: train_ops.backward, nm_mlp_backward_ex) -- the
    # transposed layers_xyz[0] is applied once to a sum over the samples instead of per sample.  `frac` counts the matrix work that is
    # EXECUTED (that layer left out of the delta term); `frac_of_reference_work` what the reference's autograd does, the basis of
    # the earlier rounds' figures
    reference_flops = flops
    share = _linear_layer1_share(Hh, rays, (NUM_COARSE, NUM_COARSE + NUM_FINE))
    delta -= 2 * Hh * Hh * share              # layers_xyz[0]^T is not applied per sample ...
    wgrad = fwd - 2 * Hh * Hh * share         # ... and layers_xyz[0]'s own weight gradient is S [W1 | b1]^T, not a product over the samples
    flops = samples * (fwd + delta + wgrad)
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"value": rays / ms * 1e3, "unit": "rays/s", "ms_per_iteration": ms, "rays_per_iteration": rays,
           "workload": "training step: 8x256 coarse+fine, 64+128 samples, perturb + noise, Adam (forward + HIP backward + step)",
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "samples_per_iteration": samples,
                        "frac_of_reference_work": reference_flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        "flop_basis": "frac: the matrix work executed (forward + delta + weight gradients; without layers_xyz[0]^T in the "
                                      "delta chain and without layers_xyz[0]'s per-sample weight-gradient product: both follow from sums "
                                      "over the samples because layer1 has no activation); frac_of_reference_work: the reference's autograd, "
                                      "which runs both per sample (the basis of rounds 4 - 5)",
                        "algorithmic_flops_per_sample": {"forward": fwd, "delta": delta, "weight_gradients": wgrad},
                        "floor_ms_at_peak": flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                        "note": "whole-iteration wall time (taping forward, delta kernel, dW kernels, encodings, compositing, Adam) "
                                "against the fp32 MFMA peak"}}
    # ---- where the iteration's time goes (HIP events around the stages of train_ops, a separate pass of `iters` iterations):
    # the three matrix stages each against the fp32 MFMA peak on their own algorithmic FLOP, everything else as milliseconds
    train_ops.profile_stages(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    torch.cuda.synchronize()
    ms_prof = (time.perf_counter() - t0) / iters * 1e3
    stages = {k: v / iters for k, v in train_ops.profile_stages(False).items()}
    work = {"taping_forward": samples * fwd, "delta": samples * delta, "weight_gradients": samples * wgrad}
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


def _linear_layer1_share(hidden, rays, samples_per_net):
    """Fraction of an iteration's samples whose backward takes layer1's gradient by linearity (the delta chain then leaves
    layers_xyz[0]^T out): every sample of the 64-wide networks' fused backward, and of the separate kernels the networks whose
    n x hidden^2 passes train_ops.LINEAR_LAYER1_MIN_WORK."""
    from nerfmeshes_amd import train_ops
    taken = [s for s in samples_per_net if hidden == 64 and (rays * s) % 128 == 0 or rays * s * hidden * hidden > train_ops.LINEAR_LAYER1_MIN_WORK]
    return sum(taken) / float(sum(samples_per_net))


def train_flops_per_sample(kw):
    """(forward, delta, weight-gradient) algorithmic FLOP per sample of a view-dependent FlexibleNeRFModel (weights only, as SURVEY
    8(d): /root/reference/src/nerf/models.py:5-58 lists the layers)."""
    Hh, L, ss = kw["hidden_size"], kw["num_layers"], kw["skip_step"]
    dx, dd = 6 * kw["num_encoding_fn_xyz"] + 3, 6 * kw["num_encoding_fn_dir"] + 3
    nskip = sum(1 for i in range(L - 1) if i % ss == 0 and i > 0 and i != L - 1)
    fwd = dx * Hh + (L - 1) * Hh * Hh + nskip * dx * Hh + Hh * Hh + Hh + (Hh + dd) * (Hh // 2) + 3 * (Hh // 2)
    delta = (L - 1) * Hh * Hh + Hh * Hh + Hh * (Hh // 2)
    return 2 * fwd, 2 * delta, 2 * fwd


def shape_train_probe(dev, name, over, rays, iters=20, replay=True):
    """One training iteration (perturb + noise, MSE per network, backward through the HIP kernels, fused Adam) of a network shape
    other than the headline's: eager, and replayed from one captured hipGraph (train_ops.GraphedStep) -- the whole iteration's
    algorithmic fp32 matrix work over its wall time against the fp32 MFMA peak."""
    from nerfmeshes_amd import models, train_ops
    from nerfmeshes_amd.nerf import CfgNode
    kw = dict(hidden_size=256, num_layers=8, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, num_coarse=64, num_fine=128,
              use_fine=True)
    kw.update(over)
    hp = S.hparams(train_perturb=True, train_noise_std=0.2, **kw)
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
            c, f = out if isinstance(out, tuple) else (out, None)
            loss = torch.nn.functional.mse_loss(c.rgb_map, target)
            if f is not None:
                loss = loss + torch.nn.functional.mse_loss(f.rgb_map, target)
            loss.backward()
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

    samples = rays * (kw["num_coarse"] + (kw["num_coarse"] + kw["num_fine"] if kw["use_fine"] else 0))
    flops = samples * sum(train_flops_per_sample(kw))
    per_net = (kw["num_coarse"],) + ((kw["num_coarse"] + kw["num_fine"],) if kw["use_fine"] else ())
    reference_flops = flops
    # layer1's and layers_xyz[0]'s gradients by linearity: two hidden x hidden products per sample (a transposed layer of the delta
    # chain, a weight-gradient product) are not executed for the networks that take the path
    flops -= samples * 4 * kw["hidden_size"] ** 2 * _linear_layer1_share(kw["hidden_size"], rays, per_net)
    frac = lambda ms: flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS  # noqa: E731
    eager = timed(build())
    out = {"workload": f"{name}: {rays} rays x {samples // rays} samples, perturb + noise, fused Adam", "rays": rays,
           "samples_per_iteration": samples, "ms_per_iteration": eager, "frac": frac(eager),
           "frac_of_reference_work": frac(eager) * reference_flops / flops,
           "floor_ms_at_peak": flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3}
    if replay:
        ms = timed(train_ops.GraphedStep(build(capturable=True)))
        out.update(ms_graph_replay=ms, frac_graph_replay=frac(ms), frac_graph_replay_of_reference_work=frac(ms) * reference_flops / flops,
                   rays_per_s_graph_replay=rays / ms * 1e3)
    return out


TINY_TRAIN = dict(hidden_size=64, num_layers=4, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4, num_coarse=32, num_fine=0,
                  use_fine=False)


def tiny_train_probe(dev, rays=8192, iters=40):
    """Config 1's training iteration (4x64, 32 coarse samples, no fine network; perturb + noise, MSE, backward through the HIP
    kernels, Adam) -- about forty launches of a few tens of microseconds: launched eagerly, and replayed from one captured hipGraph
    (train_ops.GraphedStep; the same kernels: tests/test_gpu_train.py::test_training_iteration_replays_from_a_hipgraph)."""
    r = shape_train_probe(dev, "config 1 training iteration: 4x64", TINY_TRAIN, rays, iters)
    return {"workload": r["workload"], "ms_per_iteration_eager": r["ms_per_iteration"], "frac_eager": r["frac"],
            "ms_per_iteration_graph_replay": r["ms_graph_replay"], "frac_graph_replay": r["frac_graph_replay"],
            "frac_ref_work": r["frac_graph_replay_of_reference_work"],   # counting the transposed layer the backward no longer applies per sample
            "rays_per_s_graph_replay": r["rays_per_s_graph_replay"], "floor_ms_at_peak": r["floor_ms_at_peak"],
            "note": "launch-bound: one captured hipGraph replaces ~40 launches per iteration (train_ops.GraphedStep)"}


def shapes_probe(dev):
    """The shipped configs' OTHER network shapes through a whole training iteration: nerf-colmap-fern's 8x128
    (/root/reference/config/nerf-colmap-fern.yml:115,152) and the 64-wide family of BASELINE config 1."""
    return {"8x128": shape_train_probe(dev, "8x128 coarse+fine", dict(hidden_size=128), 2048),
            "8x64": shape_train_probe(dev, "8x64 coarse+fine", dict(hidden_size=64), 2048)}
