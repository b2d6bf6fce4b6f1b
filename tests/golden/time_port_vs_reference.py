"""Is the oracle ("port", what bench.py's cpu_baseline times on the GPU box, where no reference tree exists) as fast as the
UNMODIFIED reference's NeRFModel.forward on the same host?  Same rays, same weights, same chunking (2048 rays), same torch
thread count; run in the build container (needs /root/reference):

    python tests/golden/time_port_vs_reference.py [--rays 4096] [--threads 8]   -> one JSON object (profiles/r04_port_vs_reference_cpu.json)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from make_golden import lego_rays, load_weights, mlp_kwargs  # noqa: E402
from nerfmeshes_amd import synthetic as S  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    hp = S.hparams()
    nerf, models = ref_import.load()
    m = models.NeRFModel(hp).eval()
    w = S.make_scene_weights(**mlp_kwargs(hp, "coarse"))
    load_weights(m, "model_coarse.", w)
    load_weights(m, "model_fine.", w)
    o, d, _ = lego_rays(args.rays)
    bounds = torch.tensor([2.0, 6.0])
    spec, rs = O.MLPSpec(), O.RenderSpec()

    def reference():
        with torch.no_grad():
            return torch.cat([m.forward((o, d[s:s + 2048], bounds))[1].rgb_map for s in range(0, args.rays, 2048)])

    def port():
        with torch.no_grad():
            return torch.cat([O.render(w, w, spec, spec, rs, o, d[s:s + 2048], 2.0, 6.0)[1]["rgb_map"] for s in range(0, args.rays, 2048)])

    a, b = reference(), port()
    times = {"reference": [], "port": []}
    for _ in range(args.reps):                         # interleaved: same thermal / cache conditions
        for name, fn in (("reference", reference), ("port", port)):
            t0 = time.perf_counter()
            fn()
            times[name].append(time.perf_counter() - t0)
    best = {k: min(v) for k, v in times.items()}
    print(json.dumps({
        "what": "unmodified reference NeRFModel.forward vs oracle.nerf_oracle.render (the 'port' bench.py times as cpu_baseline) on the same host",
        "workload": f"{args.rays} rays of a lego bench view, 8x256 coarse+fine, 64+128 samples, chunks of 2048", "torch_threads": args.threads,
        "host_cores": os.cpu_count(), "seconds_best_of_%d" % args.reps: best,
        "rays_per_s": {k: args.rays / v for k, v in best.items()}, "port_over_reference_time": best["port"] / best["reference"],
        "outputs_bit_identical": bool(torch.equal(a, b)), "all_times_s": times}))


if __name__ == "__main__":
    main()
