"""GPU: the generic-shape kernel family (nerfmeshes_amd/csrc/mlp_device_g.h) -- every FlexibleNeRFModel shape the reference's
constructor accepts (/root/reference/src/nerf/models.py:5-58) is served, not only the shipped configs' shapes.  Off-menu
shapes against the CPU oracle at the tuned kernels' tolerance (2e-5), through every entry point; on a menu shape the family
reproduces the tuned kernel bit for bit."""
import numpy as np
import pytest
import torch

from nerfmeshes_amd import synthetic as S
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    from nerfmeshes_amd import hip_ops
    return hip_ops


def _desc(kw):
    spec = O.MLPSpec(**kw)
    return spec, {k: getattr(spec, k) for k in ("num_layers", "hidden_size", "skip_step", "num_encoding_fn_xyz", "num_encoding_fn_dir",
                                                "include_input_xyz", "include_input_dir", "log_sampling_xyz", "log_sampling_dir", "use_viewdirs")}


def _close(got, ref, atol, what):
    got, ref = got.detach().cpu().numpy(), ref.detach().cpu().numpy()
    err = np.abs(got - ref)
    assert got.shape == ref.shape and np.all(err <= atol), f"{what}: max err {err.max():.3e} > {atol:.3e}"


# off-menu shapes: what a user's config may say that no shipped config does
OFF_MENU = [
    dict(hidden_size=512),                                                            # "a user's 8x512 ..."
    dict(num_encoding_fn_xyz=8),                                                      # "... or F_xyz = 8 checkpoint"
    dict(hidden_size=32, num_layers=4, skip_step=2, num_encoding_fn_xyz=4, num_encoding_fn_dir=2),   # the reference-script runner's network
    dict(hidden_size=48, num_layers=3, num_encoding_fn_xyz=5, num_encoding_fn_dir=3),                # odd tile count, odd argument counts
    dict(hidden_size=96, num_layers=5, skip_step=2),
    dict(hidden_size=100, num_layers=4, num_encoding_fn_xyz=7, num_encoding_fn_dir=1),               # not a multiple of 16; 50-row view layer
    dict(hidden_size=160, num_encoding_fn_xyz=12, num_encoding_fn_dir=6),
    dict(hidden_size=192, num_layers=6, skip_step=3, include_input_xyz=False),
    dict(hidden_size=256, include_input_dir=False, include_input_xyz=False),                          # no inputs in the encodings (a tuned plan serves it: zero weights on its identity step)
    dict(hidden_size=256, num_encoding_fn_xyz=15, num_encoding_fn_dir=15),                            # the longest encodings: 24 k-steps each
    dict(hidden_size=272, num_layers=4),                                                              # 17 tiles -> class 18
    dict(hidden_size=320, num_layers=5, num_encoding_fn_xyz=6),
    dict(hidden_size=384, num_layers=4, num_encoding_fn_dir=0),                                       # direction = the raw vector only; class 24: two waves per SIMD, the skip layer re-encodes
    dict(hidden_size=448, num_layers=3, num_encoding_fn_xyz=0),                                       # xyz = the raw point only; a class of 26 -- 32 tiles: output tiles split over a pair of waves
    dict(hidden_size=128, num_encoding_fn_dir=0, include_input_dir=False),                            # no direction columns at all
    dict(hidden_size=64, num_layers=2, num_encoding_fn_xyz=3, log_sampling_xyz=False, log_sampling_dir=False),
    dict(hidden_size=16, num_layers=2, num_encoding_fn_xyz=2, num_encoding_fn_dir=1),                 # one tile
    dict(hidden_size=2, num_layers=2, num_encoding_fn_xyz=1, num_encoding_fn_dir=1),                  # the narrowest network with a view layer (1 row)
    dict(hidden_size=5, num_layers=3, num_encoding_fn_xyz=2),                                         # 5 of 16 rows real; 2-row view layer
    dict(hidden_size=224, num_layers=10, skip_step=1),                                                # a skip at every layer
    dict(hidden_size=144, use_viewdirs=False, num_encoding_fn_xyz=9),
    dict(hidden_size=400, num_layers=4, use_viewdirs=False),
    # beyond the fused families (round 5): the layer-wise path (nerf_layerwise.hip) -- wider than one wavefront's registers ...
    dict(hidden_size=768, num_layers=4),
    dict(hidden_size=1024, num_layers=3, skip_step=2),
    dict(hidden_size=530, num_layers=3, num_encoding_fn_xyz=6),                                       # not a multiple of 4: 4-byte DMA pieces, 265-row view layer
    dict(hidden_size=600, num_layers=4, use_viewdirs=False),
    # an encoding longer than 24 MFMA k-steps (16 -- 31 functions): the fused instantiations with two-part encoding stages (round 5) ...
    dict(hidden_size=64, num_layers=3, num_encoding_fn_xyz=20, num_encoding_fn_dir=17),              # both encodings in two parts
    dict(hidden_size=272, num_layers=4, skip_step=2, num_encoding_fn_xyz=16),                          # 25 k-steps: the identity step alone is part two
    dict(hidden_size=448, num_layers=3, skip_step=2, num_encoding_fn_xyz=31, num_encoding_fn_dir=0),  # 48 k-steps on a split class, no direction encoding
    dict(hidden_size=100, num_layers=4, skip_step=2, num_encoding_fn_xyz=23, include_input_xyz=False, use_viewdirs=False),
    # ... and beyond 48 k-steps layer by layer
    dict(hidden_size=64, num_layers=3, num_encoding_fn_xyz=32, num_encoding_fn_dir=1),
]


def _enc_steps(spec):
    steps = [(3 * spec.num_encoding_fn_xyz + 1) // 2 + int(spec.include_input_xyz)]
    if spec.use_viewdirs:
        steps.append((3 * spec.num_encoding_fn_dir + 1) // 2 + int(spec.include_input_dir))
    return max(steps)


@pytest.mark.parametrize("kw", OFF_MENU, ids=lambda kw: "-".join(f"{k.replace('num_encoding_fn_', 'F').replace('hidden_size', 'H')}{v}" for k, v in kw.items()))
def test_off_menu_shapes_vs_oracle(ops, kw):
    spec, desc = _desc(kw)
    w = S.make_mlp_weights(29, density_gain=40.0, density_bias=1.0, **desc)
    mlp = ops.HipMLP(w, desc, "cuda")
    variant, waves = mlp.kernel_variant()
    on_menu = spec.hidden_size in (64, 128, 256) and spec.num_encoding_fn_xyz in (6, 10) and spec.num_encoding_fn_dir == 4
    beyond = spec.hidden_size > 512 or _enc_steps(spec) > 48
    assert variant == 0 if on_menu else ((variant == 2000) if beyond else (1000 <= variant < 2000 and 16 * (variant - 1000) >= spec.hidden_size)), \
        "an off-menu shape runs on the generic family, one beyond its limits layer by layer"
    g = torch.Generator().manual_seed(5)
    n = 3000
    pts = (torch.rand(n, 3, generator=g) * 2 - 1) * 4.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    ref = O.mlp_forward(w, spec, pts, dirs)
    scale = float(ref[:, 3].abs().max()) + 1.0
    got = mlp.sample_points(pts.cuda(), dirs.cuda())
    # an argument 2^k x carries 2^(k - 24) |x| of absolute error into its sine: beyond 15 functions the bar widens with it
    tol = 2e-5 * max(1.0, 2.0 ** (max(spec.num_encoding_fn_xyz, spec.num_encoding_fn_dir) - 15))
    _close(got[:, :3], ref[:, :3], tol, "rgb")
    _close(got[:, 3], ref[:, 3], tol * scale, "sigma")
    for m in (1, 16, 17, 129):                                 # ragged tails: fewer samples than a wave / a workgroup
        part = mlp.sample_points(pts[:m].cuda(), dirs[:m].cuda())
        assert torch.equal(part, got[:m]), f"n = {m}: a sample's value may not depend on the batch it is in"
    # rays (o + d t in the prologue) and the grid entry point, density-only == the full evaluation's sigma bit for bit
    rays, samples = 300, 9
    o = (torch.rand(rays, 3, generator=g) - 0.5) * 2.0
    d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1)
    t = torch.sort(torch.rand(rays, samples, generator=g) * 3.0, dim=-1).values
    ray_pts = o[:, None, :] + d[:, None, :] * t[..., None]
    ref_r = O.mlp_forward(w, spec, ray_pts.reshape(-1, 3), d[:, None, :].expand(rays, samples, 3).reshape(-1, 3))
    got_r = mlp.eval_rays(o.cuda(), d.cuda(), t.cuda()).reshape(-1, 4)
    _close(got_r[:, :3], ref_r[:, :3], tol, "rgb (rays)")
    _close(got_r[:, 3], ref_r[:, 3], tol * (float(ref_r[:, 3].abs().max()) + 1.0), "sigma (rays)")
    ax = torch.linspace(-1.2, 1.2, 11)
    full, dens = mlp.grid_query(ax, ax, ax, density_only=False), mlp.grid_query(ax, ax, ax, density_only=True)
    assert torch.equal(full[:, 3], dens)
    grid = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3)
    ref_g = O.mlp_forward(w, spec, grid, grid)
    _close(full[:, :3], ref_g[:, :3], tol, "rgb (grid)")
    _close(dens, ref_g[:, 3], tol * (float(ref_g[:, 3].abs().max()) + 1.0), "sigma (grid)")
    assert mlp.flops_per_sample() == 2 * sum(a * b for _, a, b in S.mlp_layer_shapes(**desc)), "useful FLOP only: padding is not counted"


def test_off_menu_shapes_vs_the_unmodified_reference(ops):
    """The same family against outputs of the UNMODIFIED reference's FlexibleNeRFModel for seven off-menu shapes
    (tests/golden/mlp_generic_points.npz, make_generic_golden.py; the oracle reproduces that file bit for bit)."""
    import json
    from tests.helpers import load_golden
    g = load_golden("mlp_generic_points")
    pts, dirs = torch.from_numpy(g["points"]).cuda(), torch.from_numpy(g["directions"]).cuda()
    for tag in [k[len("kwargs_"):] for k in g.files if k.startswith("kwargs_")]:
        kw = json.loads(str(g["kwargs_" + tag]))
        w = S.make_mlp_weights(int(g["seed"]), density_gain=float(g["gain"]), density_bias=float(g["bias"]), **kw)
        mlp = ops.HipMLP(w, kw, "cuda")
        assert mlp.kernel_variant()[0] >= 1000, tag
        got, ref = mlp.sample_points(pts, dirs).cpu(), torch.from_numpy(g["radiance_" + tag])
        _close(got[:, :3], ref[:, :3], 2e-5, f"{tag}: rgb vs the reference")
        _close(got[:, 3], ref[:, 3], 2e-5 * (float(ref[:, 3].abs().max()) + 1.0), f"{tag}: sigma vs the reference")


@pytest.mark.parametrize("kw", [dict(), dict(hidden_size=128), dict(hidden_size=64, num_layers=4, num_encoding_fn_xyz=6),
                                dict(hidden_size=128, num_layers=6, skip_step=2, num_encoding_fn_xyz=6), dict(use_viewdirs=False)])
def test_generic_family_reproduces_the_tuned_kernels_bit_for_bit(ops, kw):
    """NM_KERNEL_GENERIC on a menu shape: same sincosf on the same products, the same fp32 MFMA chains over the same column
    order (padded k-steps add exact zeros at the end of a chain) -> identical radiance, all four input modes."""
    spec, desc = _desc(kw)
    w = S.make_mlp_weights(3, density_gain=60.0, density_bias=2.0, **desc)
    tuned, generic = ops.HipMLP(w, desc, "cuda"), ops.HipMLP(w, desc, "cuda", force_generic=True)
    assert tuned.kernel_variant()[0] == 0 and generic.kernel_variant()[0] == 1000 + spec.hidden_size // 16
    g = torch.Generator().manual_seed(1)
    pts = ((torch.rand(4099, 3, generator=g) * 2 - 1) * 5.0).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(4099, 3, generator=g), dim=-1).cuda()
    assert torch.equal(tuned.sample_points(pts, dirs), generic.sample_points(pts, dirs))
    t = torch.sort(torch.rand(500, 12, generator=g) * 4.0 + 2.0, dim=-1).values.cuda()
    assert torch.equal(tuned.eval_rays(pts[:1], dirs[:500], t), generic.eval_rays(pts[:1], dirs[:500], t))
    ax = torch.linspace(-1.2, 1.2, 17)
    for density_only in (True, False):
        assert torch.equal(tuned.grid_query(ax, ax, ax, density_only=density_only), generic.grid_query(ax, ax, ax, density_only=density_only))
    if spec.use_viewdirs:
        view = ops.make_view(S.orbit_poses(4)[1], 40, 50, 60.0)
        near, far, u = torch.tensor([2.0]), torch.tensor([6.0]), torch.linspace(0, 1, 16)
        a = ops.render_view(tuned, tuned, view, near, far, u, torch.linspace(0, 1, 24))
        b = ops.render_view(generic, generic, view, near, far, u, torch.linspace(0, 1, 24))
        assert all(torch.equal(a[1][k], b[1][k]) for k in ("rgb_map", "depth_map", "acc_map", "weights"))


TRAIN_SHAPES = [
    dict(num_layers=4, hidden_size=100, skip_step=2, num_encoding_fn_xyz=7, num_encoding_fn_dir=1),      # 50-wide view rows: element-wise tape accesses
    dict(num_layers=3, hidden_size=48, num_encoding_fn_xyz=5, num_encoding_fn_dir=3),
    dict(num_layers=3, hidden_size=5, num_encoding_fn_xyz=2, num_encoding_fn_dir=1),                      # 5 real rows of 16, 2-row view layer
    dict(num_layers=2, hidden_size=2, num_encoding_fn_xyz=1, num_encoding_fn_dir=1),                      # the narrowest network with a view layer
    dict(num_layers=4, hidden_size=320, num_encoding_fn_xyz=6),                                           # class 20: the widest that holds the encoding in registers at two waves per SIMD
    dict(num_layers=5, hidden_size=384, skip_step=2, num_encoding_fn_xyz=6),                              # class 24: re-encoding skip layers, spilled registers outside the k-step loops
    dict(num_layers=3, hidden_size=448, num_encoding_fn_xyz=4),                                           # a split class (mlp_device_gs.h): the pair of waves exchanges activation / delta halves
    dict(num_layers=4, hidden_size=512, skip_step=2, num_encoding_fn_xyz=6),                              # the widest fused class, a skip layer
    dict(num_layers=4, hidden_size=400, skip_step=2, num_encoding_fn_xyz=5, use_viewdirs=False),          # class 26: 13 + 13 tiles, fc_out's four chains handed between the waves
    dict(num_layers=3, hidden_size=390, num_encoding_fn_xyz=3, num_encoding_fn_dir=2),                   # class 26 with a view layer of 13 tiles (7 + 6)
    dict(num_layers=5, hidden_size=80, skip_step=2, include_input_xyz=False, include_input_dir=False),
    dict(num_layers=8, hidden_size=256, num_encoding_fn_xyz=8),                                           # menu width: the hand-written dW kernels take the rows
    dict(num_layers=4, hidden_size=144, num_encoding_fn_xyz=9, use_viewdirs=False),
    dict(num_layers=3, hidden_size=128, num_encoding_fn_dir=0, include_input_dir=False),                 # no direction columns
    # the layer-wise path trains too (tape rows transposed out of its planes, delta chain on the same GEMM)
    dict(num_layers=3, hidden_size=544, skip_step=2, num_encoding_fn_xyz=6),
    dict(num_layers=3, hidden_size=96, num_encoding_fn_xyz=17, num_encoding_fn_dir=2),                  # a two-part encoding trains on the fused kernels (its taping forward)
    dict(num_layers=3, hidden_size=72, skip_step=2, num_encoding_fn_xyz=32, num_encoding_fn_dir=2),      # 49 k-steps: the layer-wise path trains it
    dict(num_layers=3, hidden_size=520, num_encoding_fn_xyz=4, use_viewdirs=False),
]


@pytest.mark.parametrize("kw", TRAIN_SHAPES, ids=lambda kw: "-".join(f"{k.replace('num_encoding_fn_', 'F').replace('hidden_size', 'H')}{v}" for k, v in kw.items()))
def test_training_off_menu_shapes_vs_autograd(ops, kw):
    """Every shape nm_mlp_create accepts also TRAINS (round 4): the generic family's taping forward (activation rows of the
    real width; radiance bit-identical to inference) and delta kernel (transposed layers on the padded width classes, ReLU'
    read off the tape), weight gradients as plain products over those rows -- all parameter gradients against fp64 autograd
    over the oracle, at the tuned shapes' tolerance (tests/test_gpu_train.py)."""
    from nerfmeshes_amd import train_ops as T
    from tests.test_gpu_train import _oracle_grads, _rays, _rel
    spec, desc = _desc(kw)
    w = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in S.make_mlp_weights(5, density_gain=30.0, density_bias=0.3, **desc).items()}
    mlp = ops.HipMLP(w, desc, "cuda")
    assert mlp.kernel_variant()[0] >= 1000
    rays, samples = 53, 11                                      # 583 samples: ragged against every workgroup size
    o, d, t = _rays(rays, samples, 7)
    grad_out = torch.randn(rays, samples, 4, generator=torch.Generator().manual_seed(1))
    rad, tape = T.forward_train(mlp, o.cuda(), d.cuda(), t.cuda())
    assert torch.equal(rad, mlp.eval_rays(o.cuda(), d.cuda(), t.cuda())), "the taping kernel must not change the output"
    assert tape["mask_h"] is None and tape["h"].shape == (spec.num_layers, rays * samples, spec.hidden_size)
    ref32, g32 = _oracle_grads(w, spec, o, d, t, grad_out, torch.float32)
    ref64, g64 = _oracle_grads(w, spec, o, d, t, grad_out, torch.float64)
    assert _rel(rad.reshape(-1, 4), ref64) < max(2e-5, 4 * _rel(ref32, ref64))
    assert float(tape["h"][1:].min()) >= 0.0
    got = T.backward(mlp, tape, rad, grad_out.cuda(), o.cuda(), d.cuda(), t.cuda())
    assert set(got) == set(g64)
    worst = {k: (_rel(got[k], ref), _rel(g32[k], ref)) for k, ref in g64.items() if ref.numel()}
    bad = {k: v for k, v in worst.items() if v[0] > max(2e-4, 20 * v[1])}
    # One ReLU whose pre-activation lies within fp32 round-off of zero may open in one evaluation and stay shut in the other:
    # that moves the weight row and the bias entry of ONE layer by one sample's contribution (seen: 8x256, F = 8 -- fc_feat's
    # pair at 4.5e-4 with every other tensor at the tight bar).  That signature, and only that, is admitted up to 1e-3
    # (the bound of the full-size reference step, tests/test_gpu_train.py::FULL_SIZE_GRAD_TOL); anything else is a mismatch.
    layers = {k.rsplit(".", 1)[0] for k in bad}
    assert len(layers) <= 1 and all(v[0] <= 1e-3 for v in bad.values()), \
        f"gradient mismatch (ours, torch-fp32) relative to fp64 autograd: {bad}"


def test_adam_trains_an_off_menu_model_through_the_module_surface(ops):
    """NeRFModel.training_step-style iterations of an 8x160, F = 12 / 6 network (no shipped config has it): loss.backward()
    reaches every parameter through the generic kernels, Adam steps lower the loss, eval follows the new weights."""
    from nerfmeshes_amd.nerf import FlexibleNeRFModel
    torch.manual_seed(0)
    net = FlexibleNeRFModel(num_layers=4, hidden_size=160, skip_step=2, num_encoding_fn_xyz=12, num_encoding_fn_dir=6).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    pts = (torch.rand(4096, 3, device="cuda") - 0.5) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(4096, 3, device="cuda"), dim=-1)
    target = torch.cat((torch.sigmoid(3.0 * pts), pts.norm(dim=-1, keepdim=True)), -1)
    losses = []
    for _ in range(30):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(net(pts, dirs), target)
        loss.backward()
        assert all(p.grad is not None for p in net.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.5 * losses[0], losses
    with torch.no_grad():
        assert abs(float(torch.nn.functional.mse_loss(net(pts, dirs), target)) - losses[-1]) < 0.5 * losses[-1]


@pytest.mark.parametrize("kw", [dict(num_layers=5, hidden_size=80, skip_step=2, num_encoding_fn_xyz=7, num_encoding_fn_dir=3),
                                dict(num_layers=3, hidden_size=576, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=3)],
                         ids=["generic-5x80", "layerwise-3x576"])
def test_render_and_module_surface_on_an_off_menu_shape(ops, kw):
    """The whole render path (nm_render_rays) and the module surface (nerf.FlexibleNeRFModel, parameter refresh after an
    in-place update, the differentiable forward) on a shape outside the menu -- one of the generic family, one beyond it
    (layer by layer, nerf_layerwise.hip: nm_mlp_refresh re-gathers its transposed weight copies too)."""
    from nerfmeshes_amd.nerf import FlexibleNeRFModel
    spec, desc = _desc(kw)
    w = S.make_mlp_weights(8, density_gain=3000.0, density_bias=100.0, **desc)
    mlp = ops.HipMLP(w, desc, "cuda")
    o, d = ops.ray_bundle(S.orbit_poses(4)[1], 800, 800, S.LEGO_FOCAL_800)
    d = d[torch.arange(0, 640000, 2500, device="cuda")].contiguous()
    cb, fb = ops.render_rays(mlp, mlp, o[None], d, torch.tensor([2.0]), torch.tensor([6.0]), torch.linspace(0, 1, 32), torch.linspace(0, 1, 48))
    rc, rf = O.render(w, w, spec, spec, O.RenderSpec(num_coarse=32, num_fine=48), o[None].cpu(), d.cpu(), 2.0, 6.0)
    _close(cb["rgb_map"], rc["rgb_map"], 1e-4, "coarse rgb_map")
    err = (fb["rgb_map"].cpu() - rf["rgb_map"]).abs().max(-1).values
    assert float((err <= 1e-4).float().mean()) >= 0.97 and float(err.max()) < 5e-2, (float(err.max()), int((err > 1e-4).sum()))
    net = FlexibleNeRFModel(**kw).cuda().eval()
    pts, dirs = torch.rand(200, 3).cuda(), torch.rand(200, 3).cuda()
    with torch.no_grad():
        a = net(pts, dirs)
        net.layer1.weight.mul_(1.5)
        b = net(pts, dirs)
        ref = O.mlp_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, spec, pts.cpu(), dirs.cpu())
    assert not torch.equal(a, b)
    _close(b[:, :3], ref[:, :3], 2e-5, "module rgb after an in-place update")
    net.train()
    net(pts, dirs).sum().backward()              # the differentiable path of an off-menu shape (gradients: the test above)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())


def test_limits_of_the_family_are_errors_with_a_reason(ops):
    from nerfmeshes_amd import _lib
    for kw, why in ((dict(hidden_size=4096, num_layers=2), "2\\^24"), (dict(hidden_size=1, num_layers=2), "0 rows"), (dict(num_encoding_fn_xyz=33), "limit is 32"),
                    (dict(num_encoding_fn_xyz=0, include_input_xyz=False), "empty")):
        spec, desc = _desc(kw)
        w = {f"{name}.{part}": np.zeros((n_out, n_in) if part == "weight" else (n_out,), dtype=np.float32)
             for name, n_out, n_in in S.mlp_layer_shapes(**desc) for part in ("weight", "bias")}
        with pytest.raises(_lib.HipLibraryError, match=why):
            ops.HipMLP(w, desc, "cuda")
    spec, desc = _desc(dict(num_encoding_fn_xyz=16, include_input_xyz=False, hidden_size=64, num_layers=2))    # 24 k-steps exactly
    w = S.make_mlp_weights(2, **desc)
    pts = torch.rand(64, 3)
    got = ops.HipMLP(w, desc, "cuda").sample_points(pts.cuda(), pts.cuda()).cpu()
    ref = O.mlp_forward(w, spec, pts, pts)
    assert float((got[:, :3] - ref[:, :3]).abs().max()) < 1e-3      # 2^15 x: the argument itself carries 2^-9 of absolute error


def test_the_deepest_network_of_the_widest_fused_class_fits_the_lds(ops):
    """The fused kernels keep every bias of the network in LDS next to the weight ring (and, for the split classes, the exchange
    slots): at the ABI's 32 layers and 512 padded columns that is 163 600 of the 163 840 bytes of a CU -- it must launch, on the
    fused class, and agree with the oracle.  (The plan lookup checks the budget and would hand a network that does not fit to the
    layer-wise path: nerf_mlp.hip find_generic_plan.)"""
    g = torch.Generator().manual_seed(9)
    pts = (torch.rand(700, 3, generator=g) * 2 - 1) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(700, 3, generator=g), dim=-1)
    spec, desc = _desc(dict(num_layers=32, hidden_size=500, skip_step=5, num_encoding_fn_xyz=4, num_encoding_fn_dir=2))
    w = S.make_mlp_weights(4, **desc)
    for k in [k for k in w if k.endswith(".weight") and k.startswith("layers_xyz")]:
        w[k] = (w[k] * np.float32(1.6)).astype(np.float32)      # keep the signal alive through 31 ReLU layers
    mlp = ops.HipMLP(w, desc, "cuda")
    assert mlp.kernel_variant()[0] == 1032
    ref = O.mlp_forward(w, spec, pts, dirs)
    got = mlp.sample_points(pts.cuda(), dirs.cuda())
    assert float(ref[:, :3].std()) > 1e-4, "the test network must not have died"
    _close(got[:, :3], ref[:, :3], 2e-5, "rgb")
    _close(got[:, 3], ref[:, 3], 2e-5 * (float(ref[:, 3].abs().max()) + 1.0), "sigma")
    assert torch.equal(mlp.grid_query(pts[:50, 0], pts[:50, 1], pts[:9, 2], density_only=True),
                       mlp.grid_query(pts[:50, 0], pts[:50, 1], pts[:9, 2], density_only=False)[:, 3])
