"""Timing of nm_weight_grad at the training shapes (fine network of a 2048-ray batch: 393 216 samples)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import hip_ops, synthetic as S, train_ops as T
kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
mlp = hip_ops.HipMLP(S.make_scene_weights(**kw), kw, "cuda")
n = 2048 * 192
out = {}
for o, s, i in ((256, 256, 256), (256, 64, 63), (128, 256, 256), (128, 64, 27)):
    d = torch.randn(n, o, device="cuda"); a = torch.randn(n, s, device="cuda")
    T._weight_grad(mlp, d, a, i); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for x, y in ev:
        x.record(); T._weight_grad(mlp, d, a, i); y.record()
    torch.cuda.synchronize()
    ms = min(x.elapsed_time(y) for x, y in ev)
    out[f"{o}x{s}"] = {"ms": ms, "tflops_on_padded_shape": 2.0 * n * o * s / (ms * 1e-3) / 1e12,
                       "GBps_operands": 4.0 * n * (o + s) / (ms * 1e-3) / 1e9}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); T._tn(d, a); t1.record(); torch.cuda.synchronize()
    out[f"{o}x{s}"]["rocblas_splitk_ms"] = t0.elapsed_time(t1)
print(json.dumps(out))
