"""GPU: the C ABI used from plain C (no Python / torch in the process): tests/cabi_smoke.c is compiled with
gcc against include/nerfmeshes_hip.h + the in-tree shared library, run, and checked against the CPU oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from nerfmeshes_amd import synthetic as S
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_through_the_c_abi(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    gcc = shutil.which("gcc") or "gcc"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = tmp_path / "cabi_smoke"
    libdir = os.path.join(ROOT, "nerfmeshes_amd", "csrc")
    subprocess.run([gcc, "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "cabi_smoke.c"),
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
                    "-L", libdir, "-lnerfmeshes_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", str(exe)], check=True)
    kw = dict(num_layers=4, hidden_size=64, skip_step=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    w = S.make_mlp_weights(31, density_gain=50.0, density_bias=1.0, **kw)
    order = ["layer1"] + [f"layers_xyz.{i}" for i in range(3)] + ["layers_dir.0", "fc_alpha", "fc_rgb", "fc_feat"]
    flat = [np.concatenate([w[n + ".weight"].ravel(), w[n + ".bias"].ravel()]) for n in order]
    flat += [(2.0 ** np.arange(6)).astype(np.float32), (2.0 ** np.arange(4)).astype(np.float32)]
    np.concatenate(flat).astype(np.float32).tofile(tmp_path / "weights.bin")
    n = 1000
    rng = np.random.default_rng(4)
    pts = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * 3
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    np.concatenate([pts.ravel(), dirs.ravel()]).tofile(tmp_path / "points.bin")
    res = subprocess.run([str(exe), str(tmp_path / "weights.bin"), str(tmp_path / "points.bin"), str(n), str(tmp_path / "out.bin")],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "abi 2" in res.stdout
    blob = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    assert blob.size == n * (4 + 4 + 64 + 4)
    got, taped, d_h0, d_last = np.split(blob, [4 * n, 8 * n, 8 * n + 64 * n])
    got, taped, d_h0, d_last = got.reshape(n, 4), taped.reshape(n, 4), d_h0.reshape(n, 64), d_last.reshape(n, 4)
    assert np.array_equal(got, taped), "nm_mlp_forward_train must reproduce nm_mlp_sample_points"
    ref = O.mlp_forward(w, O.MLPSpec(**kw), pts, dirs).numpy()
    assert np.abs(got[:, :3] - ref[:, :3]).max() < 2e-5
    assert np.abs(got[:, 3] - ref[:, 3]).max() < 2e-5 * (np.abs(ref[:, 3]).max() + 1)
    # training ABI from C: bias gradients are the column sums of the deltas it wrote (dL/d radiance = 1)
    w64 = {k: torch.as_tensor(v, dtype=torch.float64).requires_grad_(True) for k, v in w.items()}
    O.mlp_forward(w64, O.MLPSpec(**kw), torch.from_numpy(pts).double(), torch.from_numpy(dirs).double(), keep_graph=True).sum().backward()
    for name, ours in (("layer1.bias", d_h0.astype(np.float64).sum(0)), ("fc_rgb.bias", d_last[:, :3].astype(np.float64).sum(0)),
                       ("fc_alpha.bias", d_last[:, 3:].astype(np.float64).sum(0))):
        ref_g = w64[name].grad.numpy()
        assert np.abs(ours - ref_g).max() <= 2e-4 * np.abs(ref_g).max(), name
