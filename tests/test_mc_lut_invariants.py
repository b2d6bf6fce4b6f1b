"""CPU: structural facts about the MC33 tables that the parallel GPU formulation relies on."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDGE_CORNERS = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def load_luts():
    text = open(os.path.join(ROOT, "nerfmeshes_amd", "csrc", "mc_luts.h")).read()
    flat = np.array([int(x) for x in re.search(r"MC_LUT\[MC_LUT_SIZE\] = \{(.*?)\};", text, re.S).group(1)
                     .replace("\n", "").split(",") if x.strip()], dtype=np.int8)
    meta = {}
    for name, off in re.findall(r"#define MC_(\w+)_OFF (\d+)", text):
        meta[name] = dict(off=int(off))
    for name, row in re.findall(r"#define MC_(\w+)_ROW (\d+)", text):
        meta[name]["row"] = int(row)
    for name, sub in re.findall(r"#define MC_(\w+)_SUB (\d+)", text):
        meta[name]["sub"] = int(sub)
    return flat, meta


# which tilings a case may pick (rows indexed by the case's config, optionally a sub-row)
CASE_TILINGS = {1: ["TILING1"], 2: ["TILING2"], 3: ["TILING3_1", "TILING3_2"], 4: ["TILING4_1", "TILING4_2"],
                5: ["TILING5"], 6: ["TILING6_1_1", "TILING6_1_2", "TILING6_2"],
                7: ["TILING7_1", "TILING7_2", "TILING7_3", "TILING7_4_1", "TILING7_4_2"], 8: ["TILING8"],
                9: ["TILING9"], 10: ["TILING10_1_1", "TILING10_1_1_", "TILING10_1_2", "TILING10_2", "TILING10_2_"],
                11: ["TILING11"], 12: ["TILING12_1_1", "TILING12_1_1_", "TILING12_1_2", "TILING12_2", "TILING12_2_"],
                13: ["TILING13_1", "TILING13_1_", "TILING13_2", "TILING13_2_", "TILING13_3", "TILING13_3_",
                     "TILING13_4", "TILING13_5_1", "TILING13_5_2"], 14: ["TILING14"]}


def test_every_tiling_uses_exactly_the_sign_changing_edges():
    """=> the cube that CREATES an edge vertex is the first cube in scan order containing the edge,
    independent of which tilings its neighbours pick (marching_cubes.hip::owns_edge)."""
    flat, meta = load_luts()
    cases = flat[meta["CASES"]["off"]:meta["CASES"]["off"] + 512].reshape(256, 2)
    checked = 0
    for index in range(1, 255):
        kase, cfg = int(cases[index, 0]), int(cases[index, 1])
        cut = {e for e, (a, b) in enumerate(EDGE_CORNERS) if ((index >> a) & 1) != ((index >> b) & 1)}
        for name in CASE_TILINGS[kase]:
            m = meta[name]
            subs = range(m.get("sub", 1))
            for sub in subs:
                start = m["off"] + (cfg * m.get("sub", 1) + sub) * m["row"]
                row = [int(v) for v in flat[start:start + m["row"]]]
                edges = {v for v in row if 0 <= v < 12}
                assert edges == cut, (index, name, sub, sorted(edges), sorted(cut))
                assert all(-1 <= v <= 12 for v in row)
                checked += 1
    assert checked > 700


def test_lut_size_fits_the_packed_cube_code():
    flat, meta = load_luts()
    assert len(flat) < (1 << 15)          # marching_cubes.hip::pack_code keeps the offset in 15 bits
    assert max(m["row"] for n, m in meta.items() if n.startswith("TILING")) <= 36   # <= 12 triangles (4 bits)
