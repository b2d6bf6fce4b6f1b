"""The 64-wide networks' fused backward (nm_mlp_backward_fused) in a loop on a REAL tape -- the PMC passes' target: a 4x64 (default) or
8x64 network, RAYS x SAMPLES ray samples through the taping forward, then REPS backward launches; prints the mean launch time.
    python tests/tools/fused_bwd_one.py [LAYERS] [RAYS] [SAMPLES] [REPS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nerfmeshes_amd import _lib, hip_ops, synthetic as S, train_ops as T  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kw = dict(num_layers=L, hidden_size=64, skip_step=2 if L == 4 else 4, num_encoding_fn_xyz=6 if L == 4 else 10, num_encoding_fn_dir=4)
mlp = hip_ops.HipMLP({k: torch.as_tensor(v) for k, v in S.make_mlp_weights(3, density_gain=30.0, density_bias=0.3, **kw).items()}, kw, "cuda")
g = torch.Generator().manual_seed(1)
o = torch.tensor([[0.0, 0.0, 4.0]]).cuda()
d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1).cuda()
t = torch.sort(2.0 + 4.0 * torch.rand(rays, samples, generator=g), dim=-1).values.cuda()
grad = torch.randn(rays, samples, 4, generator=g).cuda()
assert _lib.load().nm_mlp_backward_fused_supported(mlp.handle, rays * samples)
rad, tape = T.forward_train(mlp, o, d, t)
for _ in range(2):
    T.backward(mlp, tape, rad, grad, o, d, t)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    T.backward(mlp, tape, rad, grad, o, d, t)
e1.record()
torch.cuda.synchronize()
alive = float((tape["h"][1:] > 0).float().mean())
print(f"{L}x64, {rays} x {samples} samples: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per backward (kernel + reduction), {alive:.2f} of the hidden activations alive")
