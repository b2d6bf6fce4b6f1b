"""GPU: the C ABI used from plain C (no Python / torch in the process): tests/cabi_smoke.c is compiled with
gcc against include/nerfmeshes_hip.h + the in-tree shared library, run, and checked against the CPU oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from nerfmeshes_amd import synthetic as S
from oracle import mc_oracle, nerf_oracle as O

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
    w = S.make_mlp_weights(31, density_gain=50.0, density_bias=6.5, **kw)    # bias: the rays of the render leg see a half-empty scene
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
    assert "abi 6" in res.stdout
    blob = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    mlp_part = n * (4 + 4 + 64 + 4)
    rest = blob[mlp_part:]
    blob = blob[:mlp_part]
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

    # ---- nm_weight_grad_ex / nm_head_grad_ex from C on n = 1000 rows (ragged): layers_xyz.0's gradient, fc_alpha's row
    Hh = 64
    gw, rest = rest[:Hh * Hh + Hh + 4 * Hh], rest[Hh * Hh + Hh + 4 * Hh:]
    for name, ours in (("layers_xyz.0.weight", gw[:Hh * Hh].reshape(Hh, Hh)), ("layers_xyz.0.bias", gw[Hh * Hh:Hh * Hh + Hh]),
                       ("fc_alpha.weight", gw[Hh * Hh + Hh:].reshape(4, Hh)[3:4])):
        ref_g = w64[name].grad.numpy()
        assert np.abs(ours - ref_g).max() <= 2e-4 * np.abs(ref_g).max(), name
    # ---- nm_render_rays from C: 256 rays, 16 + 24 samples, one network as coarse and fine
    R, G = 256, 24
    img, rest = rest[:4 * R], rest[4 * R:]
    rgb, acc = img[:3 * R].reshape(R, 3), img[3 * R:]
    _, fb = O.render(w, w, O.MLPSpec(**kw), O.MLPSpec(**kw), O.RenderSpec(num_coarse=16, num_fine=24),
                     torch.from_numpy(pts[:R]), torch.from_numpy(dirs[:R]), 0.5, 3.0)
    ok = np.abs(rgb - fb["rgb_map"].numpy()).max(-1) <= 1e-4
    assert ok.mean() >= 0.98 and np.abs(rgb - fb["rgb_map"].numpy()).max() < 5e-2, (ok.mean(), np.abs(rgb - fb["rgb_map"].numpy()).max())
    assert np.abs(acc - fb["acc_map"].numpy())[ok].max() <= 1e-4 and 0.02 < acc.mean() < 0.98
    # ---- nm_mlp_grid_query -> nm_mc_count / nm_mc_emit from C: the grid vs the oracle's, the mesh vs the C oracle on THAT grid
    grid, rest = rest[:G ** 3].reshape(G, G, G), rest[G ** 3:]
    iso, nv, nf = float(rest[0]), int(rest[1]), int(rest[2])
    rest = rest[3:]
    ref_grid = O.extract_radiance(w, O.MLPSpec(**kw), 1.2, G)[..., 3]
    assert np.abs(grid - ref_grid).max() <= 2e-5 * (np.abs(ref_grid).max() + 1)
    assert rest.size == nv * 3 + nf * 3 + nv * 3 + nv and nv > 50 and nf > 50
    v, f, nrm, val = np.split(rest, [3 * nv, 3 * nv + 3 * nf, 6 * nv + 3 * nf])
    rv, rf, rn, rval = mc_oracle.marching_cubes(np.ascontiguousarray(grid), iso)
    assert rv.shape == (nv, 3) and rf.shape == (nf, 3)
    assert v.tobytes() == rv.tobytes() and f.view(np.int32).tobytes() == rf.astype(np.int32).tobytes()
    assert nrm.tobytes() == rn.tobytes() and val.tobytes() == rval.tobytes()
    # ---- nm_mlp_backward_fused from C (ABI v6): the whole backward of the 4x64 network over the first 896 points in one call
    fused = np.fromfile(str(tmp_path / "out.bin") + ".fused", dtype=np.float32)
    m, Hh, dd = 896, 64, 27
    assert fused.size == Hh * Hh + Hh + 3 * (Hh // 2) + Hh + (Hh // 2) * (Hh + dd), "the C program did not take the fused backward"
    w64 = {k: torch.as_tensor(v, dtype=torch.float64).requires_grad_(True) for k, v in w.items()}
    O.mlp_forward(w64, O.MLPSpec(**kw), torch.from_numpy(pts[:m]).double(), torch.from_numpy(dirs[:m]).double(), keep_graph=True).sum().backward()
    parts = np.split(fused, np.cumsum([Hh * Hh, Hh, 3 * (Hh // 2), Hh]))
    for name, ours in zip(("fc_feat.weight", "layer1.bias", "fc_rgb.weight", "fc_alpha.weight", "layers_dir.0.weight"), parts):
        ref_g = w64[name].grad.numpy()
        assert np.abs(ours.reshape(ref_g.shape) - ref_g).max() <= 2e-4 * np.abs(ref_g).max(), name
