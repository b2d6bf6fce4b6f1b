"""GPU parity: nm_mc_count / nm_mc_emit vs the C oracle (itself pinned bit-for-bit to scikit-image) and the
committed scikit-image golden set.  Everything is compared BITWISE: vertices, faces (vertex numbering and
triangle order), normals, values, and the two error conditions."""
import numpy as np
import pytest
import torch

from oracle import mc_oracle
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    from nerfmeshes_amd import hip_ops
    return hip_ops


def _same(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and a.tobytes() == b.astype(a.dtype).tobytes()


def _check(ops, vol, iso, ref, tag):
    v, f, n, val = ops.marching_cubes(torch.from_numpy(np.ascontiguousarray(vol, dtype=np.float32)).cuda(), iso)
    rv, rf, rn, rval = ref
    assert f.dtype == torch.int32 and v.dtype == torch.float32
    assert _same(f, rf), f"{tag}: faces differ ({tuple(f.shape)} vs {rf.shape})"
    assert _same(v, rv), f"{tag}: vertices differ"
    assert _same(val, rval), f"{tag}: values differ"
    assert _same(n, rn), f"{tag}: normals differ (max abs {np.abs(n.cpu().numpy() - rn).max():.3e})"


def test_mc_golden_set_bitwise(ops):
    g = load_golden("mc_cases")
    for i in range(int(g["count"])):
        vol, iso = g[f"vol_{i}"], float(g[f"iso_{i}"])
        if f"err_{i}" in g.files:
            kind = {"RuntimeError": RuntimeError, "ValueError": ValueError}[str(g[f"err_{i}"]).split(":")[0]]
            with pytest.raises(kind):
                ops.marching_cubes(torch.from_numpy(vol).cuda(), iso)
            continue
        _check(ops, vol, iso, (g[f"verts_{i}"], g[f"faces_{i}"], g[f"normals_{i}"], g[f"values_{i}"]), f"golden {i}")


@pytest.mark.parametrize("shape", [(2, 2, 2), (2, 9, 3), (17, 5, 33), (64, 64, 64), (33, 130, 77), (160, 160, 160),
                                   # rows wider than one 512-voxel brick (the classify kernel's x-halo path), 16-byte loads and scalar ones,
                                   # and more planes than one march (26 > 24)
                                   (3, 9, 700), (4, 5, 515), (26, 10, 1032),
                                   # every cube cut on full-width rows for more planes than one march: the staging buffer flushes mid-march
                                   (30, 16, 512)])
@pytest.mark.parametrize("kind", ["noise", "ties", "smooth"])
def test_mc_vs_oracle_bitwise(ops, shape, kind):
    rng = np.random.default_rng(hash((shape, kind)) % (2 ** 32))
    if kind == "noise":
        vol = rng.standard_normal(shape).astype(np.float32)
        iso = 0.1
    elif kind == "ties":
        vol = rng.integers(-2, 3, shape).astype(np.float32)      # many corners exactly on the iso level
        iso = 0.0
    else:
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij"), -1)
        vol = (np.sin(3 * g[..., 0]) * np.cos(2 * g[..., 1]) + g[..., 2] ** 2 - 0.3).astype(np.float32)
        iso = float(np.float32(0.05))
    try:
        ref = mc_oracle.marching_cubes(vol, iso)
    except (RuntimeError, ValueError) as e:
        with pytest.raises(type(e)):
            ops.marching_cubes(torch.from_numpy(vol).cuda(), iso)
        return
    _check(ops, vol, iso, ref, f"{shape} {kind}")


@pytest.mark.parametrize("shape,parts", [((9, 6, 7), 2), ((33, 40, 21), 3), ((64, 64, 64), 8), ((26, 10, 1032), 4), ((5, 9, 11), 4)])
@pytest.mark.parametrize("kind", ["noise", "ties", "smooth"])
def test_mc_slabs_concatenate_to_the_whole_mesh(ops, shape, parts, kind):
    """Per-slab marching cubes (nm_mc_count_slab / nm_mc_emit_slab): the cube layers of a volume are cut into `parts` slabs,
    every slab is meshed on its own from its planes plus one ghost plane on either side, and the slabs' arrays concatenated
    in order ARE the mesh of the whole volume -- vertices, faces with their vertex numbering, normals, values, bitwise."""
    rng = np.random.default_rng(hash((shape, kind, parts)) % (2 ** 32))
    if kind == "noise":
        vol, iso = rng.standard_normal(shape).astype(np.float32), 0.1
    elif kind == "ties":
        vol, iso = rng.integers(-2, 3, shape).astype(np.float32), 0.0
    else:
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij"), -1)
        vol, iso = (np.sin(3 * g[..., 0]) * np.cos(2 * g[..., 1]) + g[..., 2] ** 2 - 0.3).astype(np.float32), float(np.float32(0.05))
    full = torch.from_numpy(vol).cuda()
    try:
        want = mc_oracle.marching_cubes(vol, iso)                      # the C oracle, not the HIP whole-grid path
    except (RuntimeError, ValueError):
        pytest.skip("no surface in this volume")
    got = _slab_mesh(ops, full, iso, parts)
    for name, a, b in zip(("vertices", "faces", "normals", "values"), got, want):
        assert _same(a, b), f"{shape} {kind} x{parts}: {name} differ from the C oracle's whole-grid mesh"
    whole = ops.marching_cubes(full, iso)                              # and the whole-grid HIP call agrees with both
    for name, a, b in zip(("vertices", "faces", "normals", "values"), got, whole):
        assert a.shape == b.shape and torch.equal(a, b), f"{shape} {kind} x{parts}: {name} differ from nm_mc_emit"


def _slab_mesh(ops, full, iso, parts):
    from nerfmeshes_amd import dist as nd
    layers = full.shape[0] - 1
    pieces, base = [], 0
    for r in range(parts):
        lo, hi = nd.split_range(layers, r, parts)
        if hi == lo:
            continue                                                   # more parts than layers: an empty slab
        below, above = int(lo > 0), int(hi < layers)
        sub = full[lo - below:hi + 1 + above].contiguous()             # planes lo - below .. hi + above
        slab = ops.marching_cubes_slab(sub, iso, lo - below, below, above)
        pieces.append(slab.emit(base - slab.ghost_vertices))
        base += slab.vertices
    return [torch.cat([p[i] for p in pieces], 0) for i in range(4)]


@pytest.mark.parametrize("tip", [3.2, 3.5, 3.9, 4.0, 4.3])
def test_mc_slab_whose_only_vertices_lie_in_the_ghost_layer(ops, tip):
    """the surface's topmost tip ends inside the last cube layer of the LOWER slab.  The upper slab then counts
    vertices in its ghost layer only -- it owns none, its output arrays have no elements (null pointers) -- and must
    return an empty piece instead of failing with 'bad argument' (which would leave the other ranks waiting in the
    triangle all-gather)."""
    n0, n1, n2 = 9, 8, 8
    z, y, x = np.meshgrid(np.arange(n0), np.arange(n1), np.arange(n2), indexing="ij")
    vol = (tip - z - 0.15 * np.hypot(y - 3.4, x - 3.6)).astype(np.float32)       # a cone whose apex sits at height `tip`
    want = mc_oracle.marching_cubes(vol, 0.0)
    full = torch.from_numpy(vol).cuda()
    lo, hi = 4, 8                                                       # upper slab: cube layers 4..7, ghost layer 3 below
    upper = ops.marching_cubes_slab(full[lo - 1:hi + 1].contiguous(), 0.0, lo - 1, 1, 0)
    if tip < 4.0:
        assert upper.ghost_vertices > 0 and upper.vertices == 0 and upper.faces == 0
    v, f, nrm, val = upper.emit(123)
    assert v.shape == (upper.vertices, 3) and f.shape == (upper.faces, 3)
    got = _slab_mesh(ops, full, 0.0, 2)
    for name, a, b in zip(("vertices", "faces", "normals", "values"), got, want):
        assert _same(a, b), f"tip {tip}: {name} differ"


def test_mc_errors(ops):
    vol = torch.ones(4, 4, 4, device="cuda")
    with pytest.raises(ValueError):
        ops.marching_cubes(vol, 2.0)               # level outside the data range
    with pytest.raises(RuntimeError):
        ops.marching_cubes(vol, 1.0)               # in range but no strictly-greater / not-greater mix
    with pytest.raises(ValueError):
        ops.marching_cubes(torch.ones(1, 4, 4, device="cuda"), 1.0)


def test_mc_properties_large(ops):
    """Size-independent checks at a size the oracle would take too long for (400^3): watertight mesh
    of a sphere (every edge shared by exactly two triangles, Euler characteristic 2) and determinism."""
    n = 400
    ax = torch.linspace(-1, 1, n, device="cuda")
    z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = (0.63 - torch.sqrt(x * x + 1.1 * y * y + 0.9 * z * z)).contiguous()
    v, f, nrm, val = ops.marching_cubes(vol, 0.0)
    v2, f2, n2, val2 = ops.marching_cubes(vol, 0.0)
    assert torch.equal(f, f2) and torch.equal(v, v2) and torch.equal(nrm, n2)
    f = f.long()
    assert int(f.min()) == 0 and int(f.max()) == v.shape[0] - 1
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    key = torch.minimum(e[:, 0], e[:, 1]) * v.shape[0] + torch.maximum(e[:, 0], e[:, 1])
    uniq, counts = torch.unique(key, return_counts=True)
    assert bool((counts == 2).all()), "closed surface: every edge belongs to exactly two triangles"
    assert v.shape[0] - uniq.numel() + f.shape[0] == 2, "Euler characteristic of a sphere"
    # outward normals ('descent': the inside is > iso, so returned normals point away from the centre)
    centre = torch.tensor([(n - 1) / 2.0] * 3, device="cuda")
    assert float(((v - centre) * nrm).sum(-1).min()) > 0
