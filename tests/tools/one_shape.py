"""One fused-MLP launch of one network shape on 2^21 points (for rocprofv3 counter runs):  python one_shape.py LAYERS HIDDEN FX"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfmeshes_amd import hip_ops, synthetic as S
layers, hidden, fx = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
n = 1 << 21
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
kw = dict(num_layers=layers, hidden_size=hidden, skip_step=4, num_encoding_fn_xyz=fx, num_encoding_fn_dir=4)
mlp = hip_ops.HipMLP(S.make_mlp_weights(3, **kw), kw, dev)
for _ in range(3):
    mlp.sample_points(pts, dirs)
torch.cuda.synchronize()
