"""PSNR-parity bookkeeping (test infrastructure, NOT product code).

north_star: "outputs match the reference PyTorch CPU path within 1e-4 PSNR on identical rays".  PSNR needs
target pixels; there is no dataset offline, so the targets are a seeded *noisy photograph of the reference
render*: tgt = clamp(ref_rgb + N(0, 0.02)), i.e. a scene the reference reproduces at ~34 dB -- the regime a
trained NeRF is scored in (/root/reference/src/eval_nerf.py:50-105), where 1e-4 dB means something:
dPSNR = 4.34 * dMSE / MSE with MSE ~ 4e-4, so 1e-4 dB corresponds to a render error of ~1e-4 rms, and the fp32
resolution of the loss (6e-8 relative) is 3e-7 dB.  (Round 1 scored against uniform-random targets: PSNR 6.8 dB,
MSE 0.21 -- a 3e-4 render error moved that by 1e-7 relative, below fp32 resolution; the check was blind.)

Both renders are scored with the reference's own loss bookkeeping (`nerf_oracle.view_loss`: per-chunk
`mse_loss`, divided by the FLOAT batch count `rays / chunk`), on ALL rays given -- nothing is filtered before
the comparison; rays above 1e-4 are counted and reported next to it.
"""
import numpy as np
import torch

from . import nerf_oracle as O

TARGET_NOISE_STD = 0.02
TARGET_SEED = 42


def noisy_targets(ref_rgb, std=TARGET_NOISE_STD, seed=TARGET_SEED):
    """(R,3) float32 targets in [0,1]: the reference render + seeded PCG64 Gaussian noise (numpy stream: stable
    across torch versions)."""
    ref = np.asarray(ref_rgb, dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(seed))
    noise = rng.standard_normal(ref.shape, dtype=np.float32) * np.float32(std)
    return torch.from_numpy(np.clip(ref + noise, 0.0, 1.0).astype(np.float32))


def psnr_parity(got_rgb, ref_rgb, chunk=2048, targets=None):
    """dict with PSNR(reference, targets), PSNR(ours, targets), |difference| in dB (the 1e-4 bar), and the direct
    measures max|d rgb|, PSNR(ours vs reference), #rays whose worst channel differs by more than 1e-4."""
    got = torch.as_tensor(np.asarray(got_rgb, dtype=np.float32))
    ref = torch.as_tensor(np.asarray(ref_rgb, dtype=np.float32))
    assert got.shape == ref.shape and got.dim() == 2 and got.shape[1] == 3, (got.shape, ref.shape)
    tgt = noisy_targets(ref) if targets is None else torch.as_tensor(targets, dtype=torch.float32)
    p_ref = float(O.mse2psnr(O.view_loss(ref, tgt, chunk)))
    p_got = float(O.mse2psnr(O.view_loss(got, tgt, chunk)))
    err = (got - ref).abs().max(-1).values
    direct = torch.nn.functional.mse_loss(got, ref)
    return {
        "psnr_ref_db": p_ref, "psnr_hip_db": p_got, "abs_dpsnr_db": abs(p_ref - p_got),
        "max_abs_drgb": float(err.max()) if err.numel() else 0.0,
        "psnr_hip_vs_ref_db": float(O.mse2psnr(direct)) if float(direct) > 0 else float("inf"),
        "rays": int(ref.shape[0]), "rays_over_1e-4": int((err > 1e-4).sum()),
        "targets": f"reference render + N(0,{TARGET_NOISE_STD}) PCG64({TARGET_SEED}), clamped to [0,1]",
        "chunk": chunk,
    }


THIN_BIN_PDF = 0.05       # a resampled depth in a coarse bin with less than this share of the pdf is ill-conditioned
SMALL_SHIFT = 5e-4        # depth units (the bench rays span 4): shifts a 1e-4 difference of the coarse weights produces


def explain_outliers(err, ref_coarse, ref_fine, t_got, rgb_on_ref_depths, tol=1e-4):
    """Every ray whose colour differs from the reference by more than `tol` must belong to a DECLARED class -- otherwise the
    conditioning argument could hide a real bug.  The classes:
      empty       the coarse pass saw almost nothing (0 < coarse acc < 5e-3): SamplePDF normalises fp32 noise;
      resampling  (a) on the reference's OWN fine depths the render agrees with the reference to round-off (5e-5): the
                  kernels are right, only the resampled depths differ; and (b) every depth that moved either moved by less
                  than SMALL_SHIFT (the coarse weights of a ray grazing a steep surface differ by ~1e-4 between two fp32
                  evaluations, the inverse CDF passes that on smoothly, and the colour next to a density step is that
                  sensitive to depth), or lies in a coarse bin holding < THIN_BIN_PDF of the pdf (a 1-ulp cdf difference
                  is amplified by 1 / pdf), or is the u = 1 sample on the cdf's final plateau (modules.py:243-246).
    Anything else -- a disagreement on identical depths, or a large move in a well-populated bin -- is `unexplained`.
    err (R,) worst-channel |d rgb|; ref_coarse / ref_fine: oracle bundles with `t`, `weights`, `acc_map`; t_got (R,Sf) the
    fine depths of the path under test; rgb_on_ref_depths (R,3) its fine render on the reference's depths.
    -> dict(rays_over_tol, empty, resampling, unexplained, unexplained_rays)."""
    err = torch.as_tensor(err, dtype=torch.float32)
    out = {"rays_over_tol": 0, "empty": 0, "resampling": 0, "unexplained": 0, "unexplained_rays": []}
    bad = torch.nonzero(err > tol).reshape(-1).tolist()
    out["rays_over_tol"] = len(bad)
    if not bad:
        return out
    acc = torch.as_tensor(ref_coarse["acc_map"])
    tc, wc = torch.as_tensor(ref_coarse["t"]), torch.as_tensor(ref_coarse["weights"])
    tf = torch.as_tensor(ref_fine["t"])
    t_got = torch.as_tensor(t_got).cpu()
    fixed = (torch.as_tensor(rgb_on_ref_depths).cpu() - torch.as_tensor(ref_fine["rgb_map"])).abs().max(-1).values
    for r in bad:
        if 0.0 < float(acc[r]) < 5e-3:
            out["empty"] += 1
            continue
        ok = float(fixed[r]) <= 5e-5
        if ok:
            w = wc[r, 1:-1] + 1e-5
            pdf = w / w.sum()
            bins = 0.5 * (tc[r, 1:] + tc[r, :-1])                      # 63 bin edges, 62 bins
            shift = (t_got[r] - tf[r]).abs()
            for j in torch.nonzero(shift > SMALL_SHIFT).reshape(-1).tolist():
                for t in (float(tf[r, j]), float(t_got[r, j])):
                    i = int(torch.searchsorted(bins, torch.tensor(t), right=True)) - 1
                    if not (i < 0 or i >= pdf.numel() or float(pdf[i]) < THIN_BIN_PDF or t >= float(bins[-2])):
                        ok = False
        if ok:
            out["resampling"] += 1
        else:
            out["unexplained"] += 1
            out["unexplained_rays"].append(int(r))
    return out


def mesh_topology(grid_hip, grid_ref, iso_hip, iso_ref, mesh_hip, mesh_ref):
    """End-to-end mesh parity of `mesh_nerf` (/root/reference/src/mesh_nerf.py:73-79): the density grid computed by the
    path under test vs the grid the CPU reference computes, each meshed at its OWN adaptive iso level by its own marching
    cubes.  On an identical grid the two marching cubes agree bitwise (tests/test_gpu_mc.py); what remains is the grid:
    a voxel whose fp32 density lies within round-off of the iso level may fall on the other side, and every such sign
    flip changes the tiling of the <= 8 cubes around it.  Reports the flips, the cubes whose 8-bit corner pattern differs
    (= the cubes whose tiling can differ), |dV|, |dF|, and checks the stated budget: isolated cubes only.
    grids: (n0,n1,n2) fp32 arrays; meshes: (vertices, faces, ...) array tuples."""
    a, b = np.asarray(grid_hip, dtype=np.float32), np.asarray(grid_ref, dtype=np.float32)
    assert a.shape == b.shape and a.ndim == 3
    inside_a, inside_b = a > np.float32(iso_hip), b > np.float32(iso_ref)        # skimage's test: value > level
    flips = inside_a != inside_b

    def patterns(m):
        m = m.astype(np.uint8)
        p = np.zeros(tuple(s - 1 for s in m.shape), dtype=np.uint8)
        for bit, (dz, dy, dx) in enumerate(((0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0))):
            p |= m[dz:m.shape[0] - 1 + dz, dy:m.shape[1] - 1 + dy, dx:m.shape[2] - 1 + dx] << bit
        return p

    pa, pb = patterns(inside_a), patterns(inside_b)
    cut = int(((pb != 0) & (pb != 255)).sum())
    differ = int((pa != pb).sum())
    scale = float(np.abs(b).max()) + 1.0
    nv_a, nf_a, nv_b, nf_b = mesh_hip[0].shape[0], mesh_hip[1].shape[0], mesh_ref[0].shape[0], mesh_ref[1].shape[0]
    out = {
        "grid": list(a.shape), "iso_hip": float(iso_hip), "iso_ref": float(iso_ref), "iso_equal": bool(np.float32(iso_hip) == np.float32(iso_ref)),
        "max_abs_dsigma_over_scale": float(np.abs(a - b).max()) / scale,
        "sign_flips_at_iso": int(flips.sum()), "voxels": int(a.size),
        "cubes_cut_by_the_surface": cut, "cubes_whose_corner_pattern_differs": differ,
        "vertices_hip": int(nv_a), "vertices_ref": int(nv_b), "abs_dV": abs(int(nv_a) - int(nv_b)),
        "faces_hip": int(nf_a), "faces_ref": int(nf_b), "abs_dF": abs(int(nf_a) - int(nf_b)),
        "budget": "sign flips <= 1e-5 of the voxels; differing cubes <= 8 per flip and <= 1e-3 of the cut cubes; |dV| <= 1e-3 V; |dF| <= 1e-3 F",
    }
    out["within_budget"] = bool(out["sign_flips_at_iso"] <= max(1, 1e-5 * a.size) and differ <= 8 * out["sign_flips_at_iso"]
                                and differ <= max(8, 1e-3 * cut) and out["abs_dV"] <= max(8, 1e-3 * nv_b) and out["abs_dF"] <= max(16, 1e-3 * nf_b))
    return out
