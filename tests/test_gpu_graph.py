"""The hot call under hipGraph capture (MI355X).

`nm_render_rays` allocates nothing and synchronises nothing (include/nerfmeshes_hip.h: caller-provided workspace and
outputs, every kernel on the caller's stream), so a chunk render can be captured once into a hipGraph and replayed --
what a launch-bound caller (the reference's 2048-ray validation chunks, eval_nerf.py:62-65) would do.  The replay must
equal the eager call bit for bit on new inputs written into the captured buffers."""
import json
import os
import time

import pytest
import torch

from nerfmeshes_amd import synthetic as S

pytestmark = pytest.mark.gpu

KW = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def test_render_rays_replays_from_a_hip_graph():
    from nerfmeshes_amd import hip_ops
    dev = torch.device("cuda")
    w = S.make_scene_weights(**KW)
    coarse, fine = hip_ops.HipMLP(w, KW, dev), hip_ops.HipMLP(w, KW, dev)
    u_c, u_f = torch.linspace(0.0, 1.0, 64).to(dev), torch.linspace(0.0, 1.0, 128).to(dev)
    near, far = torch.tensor([2.0], device=dev), torch.tensor([6.0], device=dev)
    o, d = hip_ops.ray_bundle(S.orbit_poses(1)[0], 800, 800, S.LEGO_FOCAL_800, device=dev)
    o = o[None].contiguous()
    chunk = 2048                                       # cfg.nerf.validation.chunksize of the shipped configs
    static_d = d[:chunk].clone()

    def render(dirs):
        return hip_ops.render_rays(coarse, fine, o, dirs, near, far, u_c, u_f)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on the capture stream: function attributes, workspace
        render(static_d)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        _, captured = render(static_d)
    for first in (chunk, 300000, 640000 - chunk):
        static_d.copy_(d[first:first + chunk])
        graph.replay()
        torch.cuda.synchronize()
        _, eager = render(d[first:first + chunk].contiguous())
        torch.cuda.synchronize()
        for k in eager:
            assert torch.equal(captured[k], eager[k]), (first, k)
    assert float(captured["acc_map"].max()) > 0.1      # the rays hit the scene: not a comparison of zeros

    dump = os.environ.get("NM_TEST_DUMP_DIR")
    if dump:                                           # what capture buys at the reference's chunk size (not asserted)
        def per_call(fn, reps=50):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        with open(os.path.join(dump, "graph_replay.json"), "w") as f:
            json.dump({"rays_per_call": chunk, "eager_ms": per_call(lambda: render(static_d)),
                       "graph_replay_ms": per_call(graph.replay)}, f)
