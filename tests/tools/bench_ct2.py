"""Experiment of round 4: two 16-sample column tiles per wave for the narrow networks.  A/B in ONE process
on the ablation library (python -m nerfmeshes_amd.build --ablations): the tuned kernel, the generic family (one tile per wave)
and the two-tile variant of the generic kernel (mlp_device_g2.h, NM_MLP_VARIANT=200), all three bit-identical by construction
-- which is checked -- on 2^22 points.  Prints one JSON object (profiles/r04_two_column_tiles.json)."""
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
n = 1 << 22
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)


def time_it(mlp):
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = mlp.sample_points(pts, dirs); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:]), out


res = {}
# --narrow (round 5): the width classes of 2 and 3 tiles, which have no tuned kernel -- one tile against two
NARROW = "--narrow" in sys.argv
SHAPES = ((4, 32, 4, 2, 2), (6, 48, 6, 4, 4), (8, 48, 10, 4, 4), (8, 32, 10, 4, 4)) if NARROW else \
    ((8, 128, 10, 4, 4), (6, 128, 6, 2, 4), (8, 64, 10, 4, 4), (4, 64, 6, 4, 4))
LEGS = (("tuned", None, False), ("generic_one_tile_per_wave", None, True), ("generic_two_tiles_per_wave", "200", True))
if NARROW:
    LEGS = (LEGS[1], ("generic_two_tiles_per_wave_4_waves_per_simd", "200", True), ("generic_two_tiles_per_wave_2_waves_per_simd", "201", True))
for layers, hidden, fx, skip, fd in SHAPES:
    kw = dict(num_layers=layers, hidden_size=hidden, skip_step=skip, num_encoding_fn_xyz=fx, num_encoding_fn_dir=fd)
    w = S.make_mlp_weights(3, **kw)
    row = {}
    outs = {}
    for name, env, force in LEGS:
        if env is None:
            os.environ.pop("NM_MLP_VARIANT", None)
        else:
            os.environ["NM_MLP_VARIANT"] = env
        mlp = hip_ops.HipMLP(w, kw, dev, force_generic=force)
        ms, outs[name] = time_it(mlp)
        tf = n * mlp.flops_per_sample() / (ms * 1e-3) / 1e12
        row[name] = {"ms": ms, "tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK, "kernel_variant": mlp.kernel_variant()[0]}
    first = next(iter(outs.values()))
    row["bit_identical"] = bool(all(torch.equal(first, o) for o in outs.values()))
    key = f"{layers}x{hidden} F={fx}/{fd} skip {skip}"
    res[key] = row
    print(key, {k: round(v["frac_of_fp32_mfma_peak"], 3) for k, v in row.items() if isinstance(v, dict)}, "bit-identical:", row["bit_identical"], file=sys.stderr)
os.environ.pop("NM_MLP_VARIANT", None)
print(json.dumps(res))
