"""The 64-wide networks' back-propagation in one kernel (nm_mlp_backward_fused, nerf_bwd_fused.hip: delta chain + every trunk /
view-layer weight gradient, no delta row written) against fp64 autograd over the CPU oracle -- what loss.backward() leaves in
the parameters' .grad under NeRFModel.training_step (/root/reference/src/models/model_nerf.py:88-151 through
/root/reference/src/nerf/models.py:60-80) -- and against the separate delta + weight-gradient kernels on the same tape."""
import numpy as np
import pytest
import torch

from tests.helpers import O, S

pytestmark = pytest.mark.gpu

# the tuned family's encodings (6 or 10 position functions, 4 direction functions): the handles that tape their encoding rows
SHAPES = {
    "4x64 (config 1)": dict(num_layers=4, hidden_size=64, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
    "8x64": dict(num_layers=8, hidden_size=64, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4),
    "2x64": dict(num_layers=2, hidden_size=64, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
    "3x64 no skip": dict(num_layers=3, hidden_size=64, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4),
    "5x64 skip at 3": dict(num_layers=5, hidden_size=64, skip_step=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
    "6x64 skip at 4": dict(num_layers=6, hidden_size=64, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4),
}


def _weights(kw, seed=5):
    w = S.make_mlp_weights(seed, density_gain=30.0, density_bias=0.3, **kw)
    return {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in w.items()}


def _rays(rays, samples, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([[0.2, -0.1, 3.5]]) + 0.1 * torch.randn(rays, 3, generator=g)
    d = torch.nn.functional.normalize(torch.tensor([[0.0, 0.1, -1.0]]) + 0.3 * torch.randn(rays, 3, generator=g), dim=-1)
    t = torch.sort(2.0 + 4.0 * torch.rand(rays, samples, generator=g), dim=-1).values
    return o, d, t


def _rel(got, ref):
    ref = ref.detach().to(torch.float64).cpu()
    got = got.detach().to(torch.float64).cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _oracle_grads(w, spec, o, d, t, grad_out, dtype):
    wd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in w.items() if "frequency" not in k}
    pts = O.ray_points(t.to(dtype), d.to(dtype), o.to(dtype)).reshape(-1, 3)
    dirs = d.to(dtype)[:, None, :].expand(-1, t.shape[1], -1).reshape(-1, 3)
    out = O.mlp_forward(wd, spec, pts, dirs, keep_graph=True)
    (out * grad_out.to(dtype).reshape(-1, 4)).sum().backward()
    return {k: v.grad for k, v in wd.items()}


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("rays,samples", [(8, 16), (128, 16), (1056, 64)])
def test_fused_backward_vs_fp64_autograd_and_the_separate_kernels(name, rays, samples, monkeypatch):
    """One workgroup (128 samples), a few, and more iterations than CUs (67 584 samples = 528 workgroup iterations: every workgroup
    loops, the activation-row ring and the weight ring wrap): all 2 (L + 4) gradient tensors against fp64 autograd at the training
    tests' tolerance, and against the separate kernels (same tape; a different, equally deterministic summation order)."""
    from nerfmeshes_amd import _lib, hip_ops as ops, train_ops as T
    kw = SHAPES[name]
    spec = O.MLPSpec(**kw)
    w = _weights(kw)
    mlp = ops.HipMLP(w, kw, "cuda")
    o, d, t = _rays(rays, samples, rays)
    grad_out = torch.randn(rays, samples, 4, generator=torch.Generator().manual_seed(1))
    n = rays * samples
    assert _lib.load().nm_mlp_backward_fused_supported(mlp.handle, n) == 1
    rad, tape = T.forward_train(mlp, o.cuda(), d.cuda(), t.cuda())
    if not tape["h0_taped"]:
        tape["h"][0].fill_(float("nan"))      # layer1's output is neither written nor read (gradients by linearity, either path below)
    got = T.backward(mlp, tape, rad, grad_out.cuda(), o.cuda(), d.cuda(), t.cuda())
    again = T.backward(mlp, tape, rad, grad_out.cuda(), o.cuda(), d.cuda(), t.cuda())
    monkeypatch.setenv("NM_FUSED_BACKWARD", "0")
    assert _lib.load().nm_mlp_backward_fused_supported(mlp.handle, n) == 0
    sep = T.backward(mlp, tape, rad, grad_out.cuda(), o.cuda(), d.cuda(), t.cuda())
    monkeypatch.delenv("NM_FUSED_BACKWARD")
    g32 = _oracle_grads(w, spec, o, d, t, grad_out, torch.float32)
    g64 = _oracle_grads(w, spec, o, d, t, grad_out, torch.float64)
    assert set(got) == set(g64) == set(sep)
    worst = {}
    for k, ref in g64.items():
        assert got[k].shape == ref.shape, k
        assert torch.equal(got[k], again[k]), f"{k}: the fused backward is deterministic"
        worst[k] = (_rel(got[k], ref), _rel(sep[k], ref), _rel(g32[k], ref))
    bad = {k: v for k, v in worst.items() if v[0] > max(2e-4, 20 * v[2])}
    assert not bad, f"gradient mismatch (fused, separate kernels, torch-fp32) relative to fp64 autograd: {bad}"


def test_what_the_fused_backward_does_not_serve_takes_the_separate_kernels():
    from nerfmeshes_amd import _lib, hip_ops as ops
    lib = _lib.load()
    kw = SHAPES["4x64 (config 1)"]
    mlp = ops.HipMLP(_weights(kw), kw, "cuda")
    assert lib.nm_mlp_backward_fused_supported(mlp.handle, 128 * 7) == 1
    assert lib.nm_mlp_backward_fused_supported(mlp.handle, 128 * 7 + 16) == 0          # a ragged batch
    for other in (dict(kw, hidden_size=128), dict(kw, use_viewdirs=False), dict(kw, num_layers=6, skip_step=2),   # two skip layers
                  dict(kw, num_layers=9, skip_step=4)):
        m = ops.HipMLP(_weights(other), other, "cuda")
        assert lib.nm_mlp_backward_fused_supported(m.handle, 1024) == 0, other


def test_module_surface_trains_through_the_fused_backward():
    """FlexibleNeRFModel -> loss.backward() -> Adam at config 1's shape: the loss falls and the fused path was the one that ran."""
    from nerfmeshes_amd import train_ops as T
    from nerfmeshes_amd.nerf import FlexibleNeRFModel
    torch.manual_seed(0)
    kw = SHAPES["4x64 (config 1)"]
    net = FlexibleNeRFModel(**kw).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    o, d, t = (x.cuda() for x in _rays(256, 32, 3))
    target = torch.rand(256, 32, 4, device="cuda")
    calls = []
    real = T._backward_fused
    T._backward_fused = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        losses = []
        for _ in range(40):
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(T.mlp_rays(net, o, d, t), target)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        T._backward_fused = real
    assert len(calls) == 40 and losses[-1] < 0.9 * losses[0], (len(calls), losses[::8])
