"""nm_weight_grad_batch on its own at the shapes a training iteration of the narrow shipped networks issues (8x128: eight
128x128 products of 393 216 rows in one launch, ...): ms, fraction of the fp32 MFMA peak, operand GB/s -- against what the same
launch costs inside the iteration (rocprofv3 trace) and what the dataflow can do (tests/tools/probes/dma_ring.hip).

    python tests/tools/bench_dw_batch.py [--general]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--general" in sys.argv:
    os.environ["NM_DW_GENERAL"] = "1"
import torch  # noqa: E402
from nerfmeshes_amd import hip_ops, synthetic as S, train_ops as T  # noqa: E402

kw = dict(num_layers=4, hidden_size=128, skip_step=2, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
mlp = hip_ops.HipMLP({k: torch.as_tensor(v) for k, v in S.make_mlp_weights(3, **kw).items()}, kw, "cuda")
out = {}
for o, stride, i, jobs, n in ((128, 128, 128, 8, 393216), (128, 128, 128, 8, 131072), (128, 128, 128, 1, 393216), (128, 64, 63, 2, 393216),
                              (64, 128, 128, 1, 393216), (64, 64, 64, 7, 393216), (64, 64, 64, 4, 262144), (64, 64, 63, 2, 262144),
                              (256, 256, 256, 8, 393216)):
    todo = []
    for j in range(jobs):
        d = torch.randn(n, o, device="cuda")
        a = torch.randn(n, stride, device="cuda")
        todo.append((d, a[:, :i] if i != stride else a, i, torch.empty(o, i, device="cuda"), 0, torch.empty(o, device="cuda")))
    T._weight_grad_batch(mlp, todo)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for x, y in ev:
        x.record()
        T._weight_grad_batch(mlp, todo)
        y.record()
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)[len(ev) // 2]
    out[f"{jobs} x ({o} x {i}, stride {stride}), n = {n}"] = {
        "ms": round(ms, 4), "frac_of_157.3_on_padded_shape": round(2.0 * jobs * n * o * stride / (ms * 1e-3) / 1e12 / 157.3, 3),
        "operand_TBps": round(4.0 * jobs * n * (o + stride) / (ms * 1e-3) / 1e12, 2)}
    del todo
print(json.dumps(out, indent=1))
