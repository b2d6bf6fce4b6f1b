"""Config 1's training iteration (4x64, 8192 rays x 32 samples) and the 8x64 / 8x128 iterations exactly as bench.py's `tiny.train` and
`train.shapes` objects measure them (eager and replayed from one hipGraph), on their own: the A/B target of NM_FUSED_BACKWARD=0/1.
    python tests/tools/bench_tiny_train.py [--shapes]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from benchlib import train  # noqa: E402

dev = torch.device("cuda:0")
out = {"NM_FUSED_BACKWARD": os.environ.get("NM_FUSED_BACKWARD", "(default: on)"), "tiny.train": train.tiny_train_probe(dev)}
if "--shapes" in sys.argv:
    out["train.shapes"] = train.shapes_probe(dev)
print(json.dumps(out, indent=1))
