"""CPU: the C restatement of Lewiner MC33 (oracle/mc_lewiner.c) reproduces scikit-image's compiled
`marching_cubes` BIT FOR BIT -- vertices, faces (= vertex numbering and triangle order), normals, values and
the two error conditions -- on the committed golden set (all 256 corner sign patterns, random / tied /
smooth / degenerate volumes, a NeRF density grid) and, where the black box is present, on a live fuzz."""
import os
import sys

import numpy as np
import pytest

from oracle import mc_oracle
from tests.helpers import load_golden


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.tobytes() == b.astype(a.dtype).tobytes()


def test_oracle_matches_skimage_golden_set():
    g = load_golden("mc_cases")
    n = int(g["count"])
    assert n > 650
    bad = []
    for i in range(n):
        vol, iso = g[f"vol_{i}"], float(g[f"iso_{i}"])
        if f"err_{i}" in g.files:
            kind = str(g[f"err_{i}"]).split(":")[0]
            with pytest.raises({"RuntimeError": RuntimeError, "ValueError": ValueError}[kind]):
                mc_oracle.marching_cubes(vol, iso)
            continue
        v, f, nrm, val = mc_oracle.marching_cubes(vol, iso)
        ok = (_same(f, g[f"faces_{i}"]) and _same(v, g[f"verts_{i}"]) and _same(nrm, g[f"normals_{i}"])
              and _same(val, g[f"values_{i}"]))
        if not ok:
            bad.append(i)
    assert not bad, f"cases differing from scikit-image: {bad[:10]}"


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="scikit-image black box not present")
def test_oracle_live_fuzz_against_skimage():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_mc
    try:
        cases = fuzz_mc.gen_cases(np.random.default_rng(int(os.environ.get("NM_FUZZ_SEED", "777"))), 120)[-400:]
        ref = fuzz_mc.skimage_batch(cases)
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"cannot run the scikit-image bridge here: {e}")
    stats = fuzz_mc.compare(cases, ref)
    assert stats["ok"] == len(cases), stats
