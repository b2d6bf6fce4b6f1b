"""Best-of-6 times of the taping forward and the delta kernel of off-menu network widths (generic family, mlp_device_g.h),
1024 rays x 192 samples:  python tests/tools/bench_generic_train.py [HIDDEN ...]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfmeshes_amd import _lib, hip_ops, synthetic as S, train_ops as T
from nerfmeshes_amd._lib import MlpDeltas

dev = torch.device("cuda:0"); R, SAMP = 1024, 192
g = torch.Generator(device="cuda").manual_seed(11)
t = torch.sort(2.0 + 4.0 * torch.rand(R, SAMP, device=dev, generator=g), dim=-1).values
o = torch.tensor([[0., 0., 4.]], device=dev)
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=-1)
lib = _lib.load()
out = {}
for hidden in [int(a) for a in sys.argv[1:]] or [96, 144, 160, 272, 320, 384, 448]:
    kw = dict(num_layers=8, hidden_size=hidden, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    mlp = hip_ops.HipMLP(S.make_mlp_weights(7, density_gain=30.0, **kw), kw, dev)
    rad, tape = T.forward_train(mlp, o, d, t); torch.cuda.synchronize()
    tf = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); T.forward_train(mlp, o, d, t); b.record(); torch.cuda.synchronize(); tf.append(a.elapsed_time(b))
    n = R * SAMP
    f32 = dict(dtype=torch.float32, device=dev)
    dl = [torch.empty(8, n, hidden, **f32), torch.empty(n, hidden, **f32), torch.empty(n, hidden // 2, **f32), torch.empty(n, 4, **f32)]
    ptr = lambda x: C.c_void_p(x.data_ptr())
    ct, cd = T._tape_struct(tape), MlpDeltas(*[ptr(x) for x in dl])
    grad = torch.randn(rad.shape, device=dev, generator=g)
    tb = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = lib.nm_mlp_backward(mlp.handle, n, C.byref(ct), ptr(rad), ptr(grad), C.byref(cd), None); b.record()
        torch.cuda.synchronize(); tb.append(a.elapsed_time(b))
        assert rc == 0
    variant, waves = mlp.kernel_variant()
    out[str(hidden)] = {"class": variant - 1000, "waves_per_workgroup": waves, "taping_forward_ms": min(tf),
                        "taping_forward_tflops": mlp.flops_per_sample() * n / min(tf) / 1e9, "delta_kernel_ms": min(tb)}
print(json.dumps(out))
