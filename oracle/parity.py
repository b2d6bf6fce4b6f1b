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
