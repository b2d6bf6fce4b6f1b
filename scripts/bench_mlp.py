"""Within-process interleaved A/B of the fused-MLP kernel variants (NM_MLP_VARIANT), 8x256 network.  Runs against
the SEPARATE ablation library (`python -m nerfmeshes_amd.build --ablations`; the product library holds variant 0 only
and ignores NM_MLP_VARIANT).
    python scripts/bench_mlp.py [variants ...]     e.g.  python scripts/bench_mlp.py 0 1 2 3 4 5
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import _lib, build as hip_build
if not os.path.exists(hip_build.ABLATION_LIB_PATH):
    hip_build.build(ablations=True, verbose=False)
_lib.LIB_PATH = hip_build.ABLATION_LIB_PATH           # explicit: nothing else in the package loads this library
from nerfmeshes_amd import hip_ops, synthetic as S

WRONG = set(range(11, 20)) | {22, 23, 42}     # timing-only ablations
variants = [int(v) for v in sys.argv[1:] if "," not in v] or [0]   # 0 = production, 10 = round-1 production, 9 = round-1 dataflow + scalar DMA
skews = [v for v in sys.argv[1:] if "," in v] or [None]
kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
w = S.make_scene_weights(**kw)
dev = torch.device("cuda:0")
mlps = {}
for v in variants:
    os.environ["NM_MLP_VARIANT"] = str(v)
    mlps[v] = hip_ops.HipMLP(w, kw, dev)
    print("requested variant", v, "-> bound (variant, waves/WG) =", mlps[v].kernel_variant())
n = 1 << 23
g = torch.Generator(device="cuda").manual_seed(0)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * 2.0
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
ref = None
combos = [(v, sk) for v in variants for sk in skews]
times = {c: [] for c in combos}
for rnd in range(6):
    for c in combos:
        v, sk = c
        if sk is None: os.environ.pop("NM_MLP_SKEW", None)
        else: os.environ["NM_MLP_SKEW"] = sk
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = mlps[v].sample_points(pts, dirs); b.record(); torch.cuda.synchronize()
        if rnd == 0:
            if ref is None: ref = out.clone()
            elif v not in WRONG: assert torch.equal(out, ref), f"variant {v} is not bit-identical to variant {variants[0]}"
        else:
            times[c].append(a.elapsed_time(b))
flops = n * mlps[variants[0]].flops_per_sample()
res = {}
for c in combos:
    t = sorted(times[c])
    res[str(c)] = {"min_ms": t[0], "median_ms": t[len(t) // 2], "tflops_best": flops / (t[0] * 1e-3) / 1e12,
                   "tflops_median": flops / (t[len(t) // 2] * 1e-3) / 1e12}
    print(f"variant {c}: best {res[str(c)]['tflops_best']:.1f} TF  median {res[str(c)]['tflops_median']:.1f} TF  ({t[0]:.2f} ms)")
print(json.dumps(res))
