"""Time of the numpy-exact fp32 statistics (nm_np_stats: sum / mean / var / std / min / max) on a 480^3 grid, and the check
that they equal numpy's bit for bit.  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerfmeshes_amd import hip_ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 480
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.randn(n, n, n, device="cuda", generator=g) * 9.0 + 3.0).contiguous()
st = hip_ops.np_stats(x)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record()
    hip_ops.np_stats(x)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
h = x.cpu().numpy().reshape(-1)
ref = {"sum": h.sum(), "mean": h.mean(), "var": h.var(), "std": h.std(), "min": h.min(), "max": h.max()}
same = all(np.float32(st[k]).tobytes() == np.float32(ref[k]).tobytes() for k in ref)
print(json.dumps({"elements": n ** 3, "ms_min": ms[0], "ms_median": ms[len(ms) // 2], "equals_numpy_bit_for_bit": bool(same),
                  "algorithmic_bytes": 2 * 4 * n ** 3, "GBps": 2 * 4 * n ** 3 / (ms[0] * 1e-3) / 1e9}))
