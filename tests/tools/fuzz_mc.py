"""Black-box fuzz of the C oracle against the compiled scikit-image module (build container only)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mc_oracle

PY39 = "/opt/conda/bin/python3.9"
BRIDGE = os.path.join(ROOT, "tests", "golden", "skimage_bridge.py")


def skimage_batch(cases):
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "in.npz"), **{f"vol_{i}": v for i, (v, _) in enumerate(cases)},
                 **{f"iso_{i}": np.float64(s) for i, (_, s) in enumerate(cases)})
        subprocess.run([PY39, BRIDGE, os.path.join(d, "in.npz"), os.path.join(d, "out.npz")], check=True,
                       env={**os.environ, "PYTHONPATH": ""})
        out = np.load(os.path.join(d, "out.npz"))
        return {k: out[k] for k in out.files}


def gen_cases(rng, n_small=400):
    cases = []
    # all 256 sign patterns on a single cube, several magnitude draws
    for rep in range(6):
        for idx in range(1, 255):
            signs = np.array([(idx >> k) & 1 for k in range(8)])
            mag = rng.uniform(0.05, 1.0, 8) * (1 if rep < 4 else rng.choice([0.01, 1.0, 30.0], 8))
            lew = np.where(signs == 1, mag, -mag).astype(np.float32)
            # Lewiner corner k -> (dz,dy,dx)
            pos = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]
            vol = np.zeros((2, 2, 2), np.float32)
            for k, (z, y, x) in enumerate(pos):
                vol[z, y, x] = lew[k]
            cases.append((vol, 0.0))
    for _ in range(n_small):
        shape = tuple(rng.integers(2, 7, 3))
        kind = rng.integers(0, 4)
        if kind == 0:
            vol = rng.standard_normal(shape).astype(np.float32)
        elif kind == 1:
            vol = rng.integers(-2, 3, shape).astype(np.float32)           # many exact ties with iso = 0
        elif kind == 2:
            g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij"), -1)
            vol = (np.linalg.norm(g * rng.uniform(0.5, 2, 3), axis=-1) - rng.uniform(0.3, 1.2)).astype(np.float32)
        else:
            vol = (rng.standard_normal(shape) * 10 ** rng.uniform(-3, 3)).astype(np.float32)
        lo, hi = float(vol.min()), float(vol.max())
        iso = float(rng.uniform(lo, hi)) if rng.random() < 0.7 else float(np.float32(rng.choice(vol.ravel())))
        cases.append((vol, iso))
    return cases


def compare(cases, ref):
    bad = 0
    stats = dict(verts=0, faces=0, normals=0, values=0, err=0, ok=0)
    for i, (vol, iso) in enumerate(cases):
        try:
            v, f, n, val = mc_oracle.marching_cubes(vol, iso)
            mine = None
        except Exception as e:  # noqa: BLE001
            mine = f"{type(e).__name__}: {e}"
        if f"err_{i}" in ref:
            if mine is None or str(ref[f"err_{i}"]).split(":")[0] != mine.split(":")[0]:
                stats["err"] += 1; bad += 1
                if bad <= 5: print("case", i, "skimage raised", ref[f"err_{i}"], "oracle:", mine)
            else:
                stats["ok"] += 1
            continue
        if mine is not None:
            stats["err"] += 1; bad += 1
            if bad <= 5: print("case", i, "oracle raised", mine, "but skimage returned", ref[f"verts_{i}"].shape)
            continue
        rv, rf, rn, rval = ref[f"verts_{i}"], ref[f"faces_{i}"], ref[f"normals_{i}"], ref[f"values_{i}"]
        okf = f.shape == rf.shape and np.array_equal(f, rf)
        okv = v.shape == rv.shape and np.array_equal(v.view(np.uint32), rv.astype(np.float32).view(np.uint32))
        okn = n.shape == rn.shape and np.array_equal(n.view(np.uint32), rn.astype(np.float32).view(np.uint32))
        okval = val.shape == rval.shape and np.array_equal(val, rval)
        if not okn and n.shape == rn.shape:
            okn_close = np.allclose(n, rn, atol=1e-6, equal_nan=True)
        else:
            okn_close = okn
        for name, ok in (("faces", okf), ("verts", okv), ("normals", okn), ("values", okval)):
            if not ok: stats[name] += 1
        if okf and okv and okn and okval:
            stats["ok"] += 1
        else:
            bad += 1
            if bad <= 8:
                print(f"case {i} shape {vol.shape} iso {iso}: faces {okf} ({f.shape} vs {rf.shape}) verts {okv} ({v.shape} vs {rv.shape}) normals {okn} (close {okn_close}) values {okval}")
                if vol.size == 8 and not okf:
                    print("   vol", vol.ravel().tolist()); print("   mine", f.tolist()); print("   ref ", rf.tolist())
    return stats


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    cases = gen_cases(rng, int(sys.argv[2]) if len(sys.argv) > 2 else 400)
    ref = skimage_batch(cases)
    print(compare(cases, ref), "of", len(cases))
