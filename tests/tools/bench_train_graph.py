"""The 8x256 training iteration (forward in train mode, both MSE losses, loss.backward(), Adam) replayed from ONE captured
hipGraph against the same iteration launched eagerly: ms per iteration, and -- with the randomness switched off -- the
parameters after k steps bit for bit (the graph holds the same kernels on the same arguments).

    python tests/tools/bench_train_graph.py [--iters 20]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerfmeshes_amd import models, synthetic as S  # noqa: E402
from nerfmeshes_amd.nerf import CfgNode  # noqa: E402

PEAK = 157.3
FLOPS_PER_SAMPLE = 1186816 + 1114112 + 1186816
# --small: BASELINE config 1's network and batch (4x64, 6 / 4 functions, 8192 rays x 32 coarse samples, no fine network) -- an
# iteration of about 1 ms and 40 launches, where launch gaps are a visible share
SMALL = "--small" in sys.argv
NET = dict(hidden_size=64, num_layers=4, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4, num_coarse=32, num_fine=0, use_fine=False) if SMALL else \
    dict(hidden_size=256, num_layers=8, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, num_coarse=64, num_fine=128, use_fine=True)


def build(dev, rays, stochastic, seed=0, adam=None):
    hp = S.hparams(train_perturb=stochastic, train_noise_std=0.2 if stochastic else 0.0, **NET)
    torch.manual_seed(seed)
    model = models.NeRFModel(CfgNode(hp)).to(dev)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, **(dict(capturable=True) if adam is None else adam))
    g = torch.Generator().manual_seed(1)
    dirs = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1).to(dev)
    origin = torch.tensor([[0.0, 0.0, 4.0]], device=dev)
    bounds = torch.tensor([2.0, 6.0])
    target = torch.rand(rays, 3, generator=g).to(dev)
    loss_out = torch.zeros((), device=dev)

    def iteration():
        opt.zero_grad(set_to_none=True)
        c, f = model((origin, dirs, bounds))
        loss = torch.nn.functional.mse_loss(c.rgb_map, target)
        if f is not None:
            loss = loss + torch.nn.functional.mse_loss(f.rgb_map, target)
        loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())

    return model, opt, iteration, loss_out, (dirs, target)


def capture(iteration):
    from nerfmeshes_amd import train_ops
    return train_ops.GraphedStep(iteration, warmup=3).graph      # 3 eager warm-up steps on a side stream, then the capture


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    rays = 8192 if SMALL else 2048
    out = {}
    # 1. the same kernels: deterministic configuration, 3 warm-up + k steps eagerly vs 3 warm-up + capture + k replays
    k = 5
    m_e, _, it_e, loss_e, _ = build(dev, rays, stochastic=False)
    for _ in range(3 + k):
        it_e()
    m_g, _, it_g, loss_g, _ = build(dev, rays, stochastic=False)
    graph = capture(it_g)                      # 3 warm-up iterations are real steps
    for _ in range(k):                         # (capturing records the iteration, it does not run it)
        graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(m_e.parameters(), m_g.parameters()))
    out["deterministic_parameters_bit_identical_after_steps"] = {"steps": 3 + k, "identical": bool(same),
                                                                 "loss_eager": float(loss_e), "loss_graph": float(loss_g)}
    del m_e, m_g, graph
    torch.cuda.empty_cache()
    # 2. the training configuration (perturb + noise): eager vs replay
    model, opt, iteration, loss_out, _ = build(dev, rays, stochastic=True)
    ms_eager = timed(iteration, iters)
    graph = capture(iteration)
    losses = []
    for _ in range(5):
        graph.replay()
        losses.append(float(loss_out))
    ms_graph = timed(graph.replay, iters)
    samples = rays * (64 + 64 + 128)
    del model, opt, graph
    torch.cuda.empty_cache()
    # 3. Adam's implementations, eagerly: torch's default (multi-tensor "foreach": seven launches per step) against the fused one
    ms_adam = {}
    for name, kw in (("foreach", dict()), ("fused", dict(fused=True)), ("foreach_again", dict()), ("fused_again", dict(fused=True))):
        _m, _o, it, _l, _ = build(dev, rays, stochastic=True, adam=kw)
        ms_adam[name] = round(timed(it, iters), 3)
        del _m, _o, it
        torch.cuda.empty_cache()
    out["eager_ms_per_iteration_by_adam_implementation"] = ms_adam
    for name, ms in (("eager", ms_eager), ("graph_replay", ms_graph)):
        out[name] = {"ms_per_iteration": round(ms, 3), "rays_per_s": round(rays / ms * 1e3)}
        if not SMALL:
            out[name]["frac_of_fp32_mfma_peak_whole_iteration"] = round(samples * FLOPS_PER_SAMPLE / (ms * 1e-3) / 1e12 / PEAK, 4)
    out["graph_replay"]["losses_of_five_replays"] = losses
    out["graph_replay"]["distinct_random_draws_per_replay"] = len(set(losses)) == len(losses)
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "train_graph_small.json" if SMALL else "train_graph.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
