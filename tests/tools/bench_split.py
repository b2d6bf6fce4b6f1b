"""Round 5: the width classes of 26 -- 32 tiles (hidden_size 385 -- 512) with a layer's output tiles split over a
pair of waves (mlp_device_gs.h: 8-wave workgroups, two waves per SIMD) against the one-wave-per-SIMD kernels of the same classes.
A/B in ONE process on the ablation library (NM_MLP_VARIANT=310 selects the old kernels, which only that library holds): bit-identity of the full evaluation, the
density-only grid query and a use_viewdirs = 0 network is checked, then both are timed on 2^21 points.  Prints one JSON object.

    python tests/tools/bench_split.py [--n 2097152]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfmeshes_amd import _lib, build as hip_build
if not os.path.exists(hip_build.ABLATION_LIB_PATH):
    hip_build.build(ablations=True, verbose=False)
_lib.LIB_PATH = hip_build.ABLATION_LIB_PATH            # explicit: nothing else in the package loads this library
from nerfmeshes_amd import hip_ops, synthetic as S

PEAK = 157.3
dev = torch.device("cuda:0")
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 1 << 21
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
ragged = 100003                                           # a tail tile and an odd workgroup count
axes = [torch.linspace(-1.5, 1.5, 96, device=dev) for _ in range(3)]


def time_it(fn):
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:]), out


res = {}
LEGS = [("one_wave_per_simd", "310"), ("split_two_waves_per_simd", None)]
ONLY = [int(x) for x in sys.argv[sys.argv.index("--hidden") + 1].split(",")] if "--hidden" in sys.argv else None
SHAPES = (dict(num_layers=8, hidden_size=512), dict(num_layers=8, hidden_size=480), dict(num_layers=8, hidden_size=448),
          dict(num_layers=8, hidden_size=400), dict(num_layers=4, hidden_size=400, skip_step=2, use_viewdirs=False),
          dict(num_layers=8, hidden_size=500, num_encoding_fn_xyz=15, num_encoding_fn_dir=0, include_input_dir=False))
for over in SHAPES:
    if ONLY is not None and over["hidden_size"] not in ONLY:
        continue
    kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    kw.update(over)
    w = S.make_mlp_weights(3, **kw)
    row, outs = {}, {}
    for name, env in LEGS:
        if env is None:
            os.environ.pop("NM_MLP_VARIANT", None)
        else:
            os.environ["NM_MLP_VARIANT"] = env
        mlp = hip_ops.HipMLP(w, kw, dev)
        ms, full = time_it(lambda: mlp.sample_points(pts, dirs))
        ms_d, dens = time_it(lambda: mlp.grid_query(*axes, density_only=True))
        outs[name] = (full, dens, mlp.sample_points(pts[:ragged], dirs[:ragged]))
        tf = n * mlp.flops_per_sample() / (ms * 1e-3) / 1e12
        tf_d = dens.numel() * mlp.flops_per_sample(True) / (ms_d * 1e-3) / 1e12
        row[name] = {"ms": ms, "tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK, "density_grid_frac": tf_d / PEAK,
                     "kernel_variant": mlp.kernel_variant()}
        del mlp
    a = outs["one_wave_per_simd"]
    row["bit_identical"] = bool(all(torch.equal(x, y) for name, b in outs.items() for x, y in zip(a, b)))
    row["max_abs_diff"] = max(float((x - y).abs().max()) for name, b in outs.items() for x, y in zip(a, b))
    key = " ".join(f"{k}={v}" for k, v in over.items())
    res[key] = row
    print(key, {k: (round(v["frac_of_fp32_mfma_peak"], 3), round(v["density_grid_frac"], 3)) for k, v in row.items() if isinstance(v, dict)},
          "bit-identical:", row["bit_identical"], "max diff", row["max_abs_diff"], file=sys.stderr, flush=True)
os.environ.pop("NM_MLP_VARIANT", None)
print(json.dumps(res))
