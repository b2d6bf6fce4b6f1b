"""SHA-256 of the taping forward's outputs (radiance + every tape tensor) and of the delta kernel's outputs on seeded inputs,
plus their best-of-N times: run before and after a change of the training kernels that must not change a bit.

    python tests/tools/train_checksum.py                 # product library
    python tests/tools/train_checksum.py --variant 3     # ablation library, NM_MLP_VARIANT=3: taping forward on the 3-slot dataflow
"""
import ctypes as C, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import _lib
if "--variant" in sys.argv:
    from nerfmeshes_amd import build as hip_build
    if not os.path.exists(hip_build.ABLATION_LIB_PATH):
        hip_build.build(ablations=True, verbose=False)
    _lib.LIB_PATH = hip_build.ABLATION_LIB_PATH            # explicit: nothing else in the package loads this library
    os.environ["NM_MLP_VARIANT"] = sys.argv[sys.argv.index("--variant") + 1]
from nerfmeshes_amd import hip_ops, synthetic as S, train_ops as T
from nerfmeshes_amd._lib import MlpDeltas

dev = torch.device("cuda:0")
h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
out = {}
R, SAMP = 2048, 192
g = torch.Generator(device="cuda").manual_seed(11)
t = torch.sort(2.0 + 4.0 * torch.rand(R, SAMP, device=dev, generator=g), dim=-1).values
o = torch.tensor([[0., 0., 4.]], device=dev)
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=-1)
for name, kw in (("8x256", dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)),
                 ("8x128", dict(num_layers=8, hidden_size=128, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)),
                 ("8x256_F6", dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)),
                 ("8x256_flat", dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=False))):
    w = S.make_mlp_weights(7, density_gain=30.0, **kw)
    mlp = hip_ops.HipMLP(w, kw, dev)
    rad, tape = T.forward_train(mlp, o, d, t)
    torch.cuda.synchronize()
    rec = {"radiance": h(rad)}
    rec.update({k: h(v) for k, v in tape.items() if v is not None})
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); T.forward_train(mlp, o, d, t); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    rec["forward_train_ms"] = min(ts)
    grad = torch.randn(rad.shape, device=dev, generator=g)
    L, H, n = kw["num_layers"], kw["hidden_size"], R * SAMP
    flat = not kw.get("use_viewdirs", True)
    f32 = dict(dtype=torch.float32, device=dev)
    deltas = {"h": torch.empty(L, n, H, **f32), "feat": None if flat else torch.empty(n, H, **f32),
              "v": None if flat else torch.empty(n, H // 2, **f32), "last": torch.empty(n, 4, **f32)}
    ptr = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    ct = T._tape_struct(tape)
    cd = MlpDeltas(ptr(deltas["h"]), ptr(deltas["feat"]), ptr(deltas["v"]), ptr(deltas["last"]))
    lib = _lib.load()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.nm_mlp_backward(mlp.handle, n, C.byref(ct), ptr(rad), ptr(grad), C.byref(cd), None)
        b.record(); torch.cuda.synchronize()
        assert rc == 0, _lib.last_error() if hasattr(_lib, "last_error") else rc
        ts.append(a.elapsed_time(b))
    rec["backward_ms"] = min(ts)
    rec.update({"delta_" + k: h(v) for k, v in deltas.items() if v is not None})
    out[name] = rec
print(json.dumps(out))
