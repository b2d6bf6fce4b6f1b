"""Shared constants and timing helpers of bench.py's probes (the workload of BASELINE.json configs[1] and the guide's peaks)."""
import json
import os
import socket
import time

import torch

from nerfmeshes_amd import hip_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def coarse_flops_per_sample():
    kw = MLP_KW
    H_, L_, dx, dd = kw["hidden_size"], kw["num_layers"], 6 * kw["num_encoding_fn_xyz"] + 3, 6 * kw["num_encoding_fn_dir"] + 3
    nskip = sum(1 for i in range(L_ - 1) if i % kw["skip_step"] == 0 and i > 0 and i != L_ - 1)
    return 2 * (dx * H_ + (L_ - 1) * H_ * H_ + nskip * dx * H_ + H_ * H_ + H_ + (H_ + dd) * (H_ // 2) + 3 * (H_ // 2))
