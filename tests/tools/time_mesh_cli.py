"""Config 4 end to end through the script entry point: `mesh_nerf --res 480 --iso-level 32 --limit 1.2
--view-disparity-max-bound 1e0` on a seeded synthetic checkpoint in the Lightning layout (density grid on the GPU,
marching cubes, per-vertex appearance re-query, OBJ text).  Prints one JSON object with the wall time per stage."""
import contextlib
import io
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nerfmeshes_amd import mesh_nerf  # noqa: E402
from nerfmeshes_amd.nerf import nerf_helpers  # noqa: E402

tmp = tempfile.mkdtemp()
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_checkpoint.py"), "--out", tmp], check=True,
               capture_output=True)
argv = ["--log-checkpoint", os.path.join(tmp, "synthetic", "default", "version_0"), "--res", str(int(os.environ.get("RES", "480"))),
        "--iso-level", "32", "--limit", "1.2", "--view-disparity-max-bound", "1e0", "--save-dir", tmp]
stages = {}
orig_export, orig_geometry = nerf_helpers.export_obj, mesh_nerf.extract_geometry


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        stages[name] = stages.get(name, 0.0) + time.perf_counter() - t0
        return out
    return wrapper


mesh_nerf.export_obj = timed("obj_text_s", mesh_nerf.export_obj)
mesh_nerf.extract_geometry = timed("density_grid_and_marching_cubes_s", mesh_nerf.extract_geometry)
for run in range(2):                      # the second run is the warm one
    stages.clear()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        mesh_nerf.main(argv)
    total = time.perf_counter() - t0
obj = [f for f in os.listdir(tmp) if f.endswith(".obj")]
size = os.path.getsize(os.path.join(tmp, obj[0])) if obj else 0
stages["appearance_requery_and_rest_s"] = total - sum(stages.values())
print(json.dumps({"argv": " ".join(argv[2:-2]), "total_s (model load + grid + MC + appearance + OBJ)": total, **stages,
                  "obj_bytes": size}))
