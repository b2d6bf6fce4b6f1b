"""Golden vectors for marching cubes from the compiled scikit-image (0.18.3, /opt/conda/bin/python3.9) --
the black box behind /root/reference/src/mesh_nerf.py:79.  Build container only.

    python tests/golden/make_mc_golden.py      # writes tests/golden/mc_cases.npz (+ prints a fuzz summary)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import fuzz_mc  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    cases = fuzz_mc.gen_cases(rng, 160)
    # keep: every 256-pattern cube once (first repetition + the extreme-magnitude repetition), 160 random volumes
    keep = list(range(0, 254)) + list(range(254 * 4, 254 * 5)) + list(range(254 * 6, len(cases)))
    cases = [cases[i] for i in keep]
    # hand-made degenerate cubes found while pinning the oracle (a == b == 0 in the case-4 test, exact zeros)
    pos = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]
    for lew in ([-1, 3, 2, 1, 1, 1, -1, 4], [1, -3, -2, -1, -1, -1, 1, -4], [-1, 4, 2, 2, 1, 2, -1, 5],
                [-1, 1, 0, 1, 1, -1, 0, -1], [-1, 1, -1, 1, 0, -1, 1, -1], [0, -2, -1, -1, -1, -1, 0, -2]):
        vol = np.zeros((2, 2, 2), np.float32)
        for k, p in enumerate(pos):
            vol[p] = lew[k]
        cases.append((vol, 0.0 if lew[0] != 0 else -1.1355386678545951))
    # a NeRF density grid (the res-20 golden of the unmodified reference), at its own iso level and at 32
    g = np.load(os.path.join(HERE, "grid_8x256_res20.npz"))
    den = np.ascontiguousarray(g["radiance"][..., 3])
    cases.append((den, float(g["iso"])))
    cases.append((den, 32.0))
    # smooth analytic shapes (cf. skimage's own tests: ellipsoid, double torus)
    z, y, x = np.meshgrid(*[np.linspace(-1.5, 1.5, 24)] * 3, indexing="ij")
    cases.append(((x ** 2 / 1.0 + y ** 2 / 0.6 + z ** 2 / 0.3).astype(np.float32), 1.0))
    cases.append((((x * (x - 1) ** 2 * (x - 2) + y ** 2) ** 2 + z ** 2).astype(np.float32), 0.05))
    # no-surface and out-of-range errors
    cases.append((np.ones((3, 3, 3), np.float32), 1.0))
    cases.append((np.ones((3, 3, 3), np.float32), 2.0))
    ref = fuzz_mc.skimage_batch(cases)
    stats = fuzz_mc.compare(cases, ref)
    print("oracle vs skimage on the golden set:", stats, "of", len(cases))
    out = {"count": len(cases)}
    for i, (vol, iso) in enumerate(cases):
        out[f"vol_{i}"], out[f"iso_{i}"] = vol, np.float64(iso)
        for k in ("verts", "faces", "normals", "values", "err"):
            if f"{k}_{i}" in ref:
                out[f"{k}_{i}"] = ref[f"{k}_{i}"]
    np.savez_compressed(os.path.join(HERE, "mc_cases.npz"), **out)
    print("wrote mc_cases.npz", os.path.getsize(os.path.join(HERE, "mc_cases.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
