"""Host-side data feed: DataBundle round trips and the per-image `.data` ray cache
(datasets.py:136-283) -- CPU only."""
import os

import pytest
import torch

from nerfmeshes_amd import synthetic as S
from nerfmeshes_amd.data import CachedRayDataset, DataBundle, DatasetType
from nerfmeshes_amd.models.model_helpers import nest_dict
from nerfmeshes_amd.nerf import CfgNode


def _cfg(tmp_path, rays=64):
    flat = S.hparams()
    flat.update({"dataset.caching.cache_dir": str(tmp_path / "cache"), "nerf.train.num_random_rays": rays})
    return CfgNode(nest_dict(flat, sep="."))


def _view(h, w, k):
    g = torch.Generator().manual_seed(k)
    dirs = torch.randn(h, w, 3, generator=g)
    return DataBundle(ray_origins=torch.tensor([0.0, 1.0, float(k)]), ray_directions=dirs, ray_targets=dirs * 0.5 + 0.25,
                      ray_bounds=torch.tensor([2.0, 6.0]), size=1, hwf=(h, w, 55.5))


def test_databundle_round_trip_and_ray_batch_shapes():
    b = _view(6, 9, 0)
    d = b.serialize(["ray_origins", "ray_directions", "ray_targets", "ray_bounds", "target_depth", "size", "hwf"])
    assert set(d) == {"ray_origins", "ray_directions", "ray_targets", "ray_bounds", "size", "hwf"}     # None is dropped
    loader_batch = {k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in d.items()}          # DataLoader batch dim
    rb = DataBundle.deserialize(loader_batch).to("cpu").to_ray_batch()
    assert rb.ray_origins.shape == (1, 3) and rb.ray_directions.shape == (54, 3) and rb.ray_bounds.shape == (2,)
    assert rb.ray_targets.shape == (54, 3) and rb.target_depth is None
    o, t = rb["ray_origins", "ray_targets"]
    assert o is rb.ray_origins and t is rb.ray_targets


def test_ray_cache_files_and_random_sampling(tmp_path):
    cfg = _cfg(tmp_path)
    writer = CachedRayDataset(cfg, DatasetType.TRAIN)
    assert len(writer) == 0
    for k in range(3):
        writer.write_view(_view(10, 14, k), k)
    files = sorted(p.name for p in (tmp_path / "cache" / "train").iterdir())
    assert files == ["0000.data", "0001.data", "0002.data"]
    raw = torch.load(tmp_path / "cache" / "train" / "0001.data", weights_only=False)      # a plain dict, as the reference writes
    assert isinstance(raw, dict) and raw["ray_directions"].shape == (10, 14, 3) and tuple(raw["hwf"]) == (10, 14, 55.5)

    train = CachedRayDataset(cfg, DatasetType.TRAIN)
    assert len(train) == 3 and train.coords.shape == (140, 2)
    assert int(train.coords[:, 0].max()) == 9 and int(train.coords[:, 1].max()) == 13
    torch.manual_seed(0)
    item = train[2]
    assert item["ray_directions"].shape == (64, 3) and item["ray_targets"].shape == (64, 3)
    assert torch.equal(item["ray_targets"], item["ray_directions"] * 0.5 + 0.25)           # pixels stay paired
    assert torch.equal(item["ray_origins"], torch.tensor([0.0, 1.0, 2.0]))                # shared origin untouched
    flat = raw_dirs = torch.load(tmp_path / "cache" / "train" / "0002.data", weights_only=False)["ray_directions"].reshape(-1, 3)
    assert all(bool((flat == row).all(-1).any()) for row in item["ray_directions"][:8])    # drawn from that image
    assert raw_dirs is flat

    for k in range(2):
        CachedRayDataset(cfg, DatasetType.VALIDATION).write_view(_view(10, 14, 10 + k), k)
    val = CachedRayDataset(cfg, DatasetType.VALIDATION)
    assert len(val) == 2 and val[0]["ray_directions"].shape == (10, 14, 3)                # whole image, no sampling


def test_obj_writer_prints_numbers_exactly_like_python(tmp_path):
    """(f)-1: nm_export_obj formats every float as Python's "{}".format(tensor_element) does (repr of the widened
    double) -- specials, both notation switches (1e-4, 1e16), denormals, and 60 000 random fp32 bit patterns."""
    import struct
    import numpy as np
    from nerfmeshes_amd.nerf import export_obj
    vals = [0.0, -0.0, 1.0, -1.5, 1e-4, 9.999e-5, 1e-5, 123456.0, 1e16, 9999999827968.0, 3.4e38, 1e-38, 1.4e-45,
            float("inf"), float("-inf"), float("nan"), 16777216.0, 0.1, 100.0, 1e15, 1e17, 9.9999998e15]
    rng = np.random.default_rng(12)
    vals += [struct.unpack("f", struct.pack("I", int(b)))[0] for b in rng.integers(0, 2 ** 32, 60000, dtype=np.uint64)]
    vals += [0.0] * (-len(vals) % 3)
    arr = np.array(vals, dtype=np.float32).reshape(-1, 3)
    tri = rng.integers(0, arr.shape[0], (50, 3)).astype(np.int32)
    path = tmp_path / "fuzz.obj"
    export_obj(arr, tri, arr[:7], arr[:11], str(path))
    lines = path.read_text().split("\n")
    n = arr.shape[0]
    fmt = lambda row: " ".join(repr(float(x)) for x in row)  # noqa: E731
    for i in range(n):
        want = "v " + fmt(arr[i]) + (" " + fmt(arr[i]) if i < 7 else "")
        assert lines[i] == want, (i, want, lines[i])
    assert lines[n:n + 11] == ["vn " + fmt(arr[i]) for i in range(11)]
    assert lines[n + 11:n + 61] == ["f " + " ".join(f"{a + 1}//{a + 1}" for a in t) for t in tri.tolist()]
    assert lines[n + 61:] == [""]


def test_the_training_wrappers_contain_no_library_gemm():
    """Every weight / bias gradient is a hand-written kernel (nm_weight_grad_ex / nm_head_grad_ex): the host wrappers of the
    training path must not call a matrix product or a reduction of torch's (rocBLAS / at::native kernels)."""
    import inspect
    import re
    from nerfmeshes_amd import train_ops
    src = inspect.getsource(train_ops)
    code = "\n".join(line.split("#")[0] for line in src.split("\n"))
    code = re.sub(r'"""[\s\S]*?"""', "", code)
    for banned in ("torch.bmm", "torch.mm", "torch.matmul", " @ ", ".sum(", "torch.einsum", "torch.addmm", ".t()"):
        assert banned not in code, f"train_ops uses {banned!r}"
    names = train_ops.param_names(8)
    assert len(names) == 2 + 2 * 7 + 8 and names[0] == "layer1.weight" and names[-1] == "fc_rgb.bias"


def test_reads_a_cache_written_by_the_reference(tmp_path):
    """(f)-4: tests/golden/ref_cache/{train,val}/NNNN.data were written by the UNMODIFIED reference's CachingDataset
    (cache_dataset -> get_ray_bundle [-> ndc] -> save_dataset; tests/golden/make_golden.py --ref-cache).
    CachedRayDataset reads them as they are, and __getitem__ under the same torch seed returns what the reference's
    __getitem__ returned (recorded in getitem_seed7.npz): the same random ray subset, pixel for pixel."""
    import os
    import numpy as np
    from oracle import nerf_oracle as O
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cache")
    rec = np.load(os.path.join(root, "getitem_seed7.npz"))
    flat = S.hparams(num_coarse=8, num_fine=8)
    flat.update({"dataset.caching.cache_dir": root, "dataset.caching.use_caching": True, "nerf.train.num_random_rays": 40})
    train = CachedRayDataset(CfgNode(nest_dict(flat, sep=".")), DatasetType.TRAIN)
    assert len(train) == 2 and train.coords.shape == (12 * 16, 2)
    raw = torch.load(train.paths[1], weights_only=False)
    assert set(raw) == {"ray_origins", "ray_directions", "ray_targets", "ray_bounds", "size", "hwf"}
    o, d = O.get_ray_bundle(12, 16, 20.0, S.orbit_poses(5)[2])
    assert torch.equal(raw["ray_directions"], d) and torch.equal(raw["ray_origins"], o)       # the reference's rays
    torch.manual_seed(7)
    item = train[1]
    for k in ("ray_origins", "ray_directions", "ray_targets", "ray_bounds"):
        assert np.array_equal(item[k].numpy(), rec["train." + k]), k
    assert item["ray_directions"].shape == (40, 3) and tuple(item["hwf"]) == (12, 16, 20.0)
    # validation split cached with use_ndc=True: per-pixel NDC origins, whole image, no sub-sampling
    flat["dataset.use_ndc"] = True
    val = CachedRayDataset(CfgNode(nest_dict(flat, sep=".")), DatasetType.VALIDATION)
    item = val[1]
    no, nd = O.ndc_rays(12, 16, 20.0, 1.0, o[None, None, :], d)
    assert torch.equal(item["ray_origins"], no) and torch.equal(item["ray_directions"], nd)
    for k in ("ray_origins", "ray_directions", "ray_targets"):
        assert np.array_equal(item[k].numpy(), rec["val." + k]), k
    # (a reference quirk preserved in the files: DataBundle.__getitem__ indexes EVERY tensor whose first dimension equals
    # the image count, so with exactly two images the (2,) ray_bounds is cut down to one scalar -- data_helpers.py:98-101)
    assert item["ray_bounds"].shape == () and np.array_equal(item["ray_bounds"].numpy(), rec["val.ray_bounds"])


def test_tie_order_auto_falls_back_to_stable_beyond_the_reference_kernels_limit(monkeypatch):
    """`tree.tie_order = "auto"` wants the reference's voxel-id order (nm_buff_intersect_ex, NM_TIES_REFERENCE),
    whose kernel holds at most 8192 voxels: beyond that "auto" continues in the stable order and says so ONCE instead of
    failing in the middle of a training run; an explicit "reference" is passed through (and would raise in the library)."""
    import warnings
    from nerfmeshes_amd import hip_ops
    from nerfmeshes_amd.nerf.tree import TreeSampling
    seen = []
    monkeypatch.setattr(hip_ops, "buff_intersect", lambda voxels, o, d, near, far, n, ids: seen.append(ids) or "ok")
    tree = TreeSampling.__new__(TreeSampling)          # no device work: only the order selection is under test
    tree.config = CfgNode({"tree": {"use_random_sampling": False}})
    tree.tie_order, tree.training = "auto", True
    o, d = torch.zeros(1, 3), torch.ones(4, 3)
    tree.voxels = torch.zeros(hip_ops.BUFF_REFERENCE_MAX_VOXELS, 2, 3)
    assert tree.batch_ray_voxel_intersect(o, d, 2.0, 6.0, 8) == "ok" and seen == ["reference"]
    tree.voxels = torch.zeros(hip_ops.BUFF_REFERENCE_MAX_VOXELS + 1, 2, 3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tree.batch_ray_voxel_intersect(o, d, 2.0, 6.0, 8)
        tree.batch_ray_voxel_intersect(o, d, 2.0, 6.0, 8)
    assert seen[1:] == ["stable", "stable"] and tree.tie_order == "stable"
    assert len([x for x in w if "tie_order" in str(x.message)]) == 1, "said once"
    tree.training = False
    tree.tie_order = "auto"
    tree.batch_ray_voxel_intersect(o, d, 2.0, 6.0, 8)
    assert seen[-1] == "stable"                        # still beyond the limit: the fallback, in evaluation as in training
    tree.voxels = torch.zeros(64, 2, 3)
    tree.tie_order = "auto"
    tree.batch_ray_voxel_intersect(o, d, 2.0, 6.0, 8)
    assert seen[-1] == "reference"                     # round 5: the reference's own ids by default in evaluation too
    tree.tie_order = "reference"
    tree.batch_ray_voxel_intersect(o, d, 2.0, 6.0, 8)
    assert seen[-1] == "reference"                     # explicit request: not second-guessed


def test_an_optimizer_step_marks_the_modules_whose_parameters_it_holds_and_no_others():
    """FlexibleNeRFModel.hip() under the "key" guard re-packs its device copy when autograd's version counters moved OR the module's
    generation did: the latter advances when an optimizer that HOLDS one of the module's parameters steps (one scoped post-step
    hook: torch's fused optimizers update tensors without touching the counters); optimizers of unrelated modules move nothing,
    and train_ops.generation() -- every module at once -- moves only on a replayed graph."""
    import torch
    from nerfmeshes_amd import train_ops
    from nerfmeshes_amd.nerf.models import FlexibleNeRFModel
    kw = dict(num_layers=2, hidden_size=16, skip_step=4, num_encoding_fn_xyz=2, num_encoding_fn_dir=1)
    mine, other = FlexibleNeRFModel(**kw), FlexibleNeRFModel(**kw)
    for m in (mine, other):
        train_ops.register_owner(m, m.parameters())      # what hip() does when it builds the handle (no GPU here)
    unrelated = torch.nn.Linear(3, 2)
    everything = train_ops.generation()

    def step(opt, params):
        for p in params:
            p.grad = torch.ones_like(p)
        opt.step()

    for n, opt in enumerate((torch.optim.SGD(mine.parameters(), lr=0.1), torch.optim.Adam(mine.parameters(), lr=0.1, foreach=False),
                             train_ops.make_optimizer("Adam", mine.parameters(), 0.1)), 1):
        step(opt, list(mine.parameters()))
        assert (mine._generation, other._generation) == (n, 0)
    step(torch.optim.SGD(unrelated.parameters(), lr=0.1), list(unrelated.parameters()))
    assert (mine._generation, other._generation) == (3, 0)
    step(torch.optim.SGD(list(other.layer1.parameters()) + list(unrelated.parameters()), lr=0.1), list(other.layer1.parameters()))
    assert (mine._generation, other._generation) == (3, 1), "one step marks a module once, whatever share of its parameters it holds"
    assert train_ops.generation() == everything
    train_ops.parameters_changed()
    assert train_ops.generation() == everything + 1
    for mode in ("always", "key", "check"):
        mine.weights_guard = mode
        assert train_ops.guard_mode(mine) == mode
    mine.weights_guard = "sometimes"
    try:
        train_ops.guard_mode(mine)
        raise AssertionError("an unknown guard must be refused")
    except ValueError:
        pass
    lin = torch.nn.Linear(3, 2)
    assert "fused" not in train_ops.make_optimizer("Adam", lin.parameters(), 0.1).defaults or \
        not train_ops.make_optimizer("Adam", lin.parameters(), 0.1).defaults["fused"], "host parameters: torch's default implementation"
    assert train_ops.make_optimizer("SGD", lin.parameters(), 0.1, momentum=0.9).defaults["momentum"] == 0.9


# ---- LLFF / COLMAP scenes (ColmapDataset.load_dataset: /root/reference/src/data/datasets.py:325-357 over loaders/load_llff.py)
def _llff_scene(tmp_path):
    import numpy as np
    from PIL import Image
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llff_scene.npz"))
    np.save(tmp_path / "poses_bounds.npy", gold["poses_bounds"])
    views, fh, fw, _ = gold["images_full_shape"]
    for name, stack in (("images", np.zeros((views, fh, fw, 3), np.uint8)), ("images_4", gold["images_4"])):
        (tmp_path / name).mkdir()
        for i, img in enumerate(stack):
            Image.fromarray(img).save(tmp_path / name / f"view_{i:03d}.png")
    return gold


@pytest.mark.parametrize("mode", ["forward", "spherify"])
def test_llff_reader_equals_the_unmodified_reference(tmp_path, mode):
    """poses_bounds.npy + images_4/ -> images, recentred (spherified) poses, rescaled bounds, the 120 render poses and the hold-out
    view, against what the unmodified reference's load_llff_data returned for the same folder (tests/golden/make_llff_golden.py)."""
    import numpy as np
    from nerfmeshes_amd.data.loaders.load_llff import load_llff_data
    gold = _llff_scene(tmp_path)
    images, poses, bounds, render_poses, i_test = load_llff_data(str(tmp_path), factor=4, spherify_poses=mode == "spherify")
    assert i_test == int(gold[f"{mode}_i_test"])
    for name, got in (("images", images), ("poses", poses), ("bounds", bounds), ("render_poses", render_poses)):
        want = gold[f"{mode}_{name}"]
        assert got.shape == want.shape and got.dtype == want.dtype, name
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6, err_msg=name)
    assert np.array_equal(images, gold[f"{mode}_images"])                    # decoding + / 255 is exact
    assert tuple(poses[0, :, 4]) == (6.0, 8.0, 30.0)                         # hwf follows the down-scaled images


def test_llff_reader_reports_what_is_missing(tmp_path):
    from nerfmeshes_amd.data.loaders.load_llff import load_llff_data
    _llff_scene(tmp_path)
    with pytest.raises(FileNotFoundError, match="mogrify"):
        load_llff_data(str(tmp_path), factor=8)                               # images_8/ was never made
    os.remove(tmp_path / "images_4" / "view_000.png")
    with pytest.raises(ValueError, match="8 images .* 9 poses"):
        load_llff_data(str(tmp_path), factor=4)


def test_colmap_dataset_splits_an_llff_scene_as_the_reference_does(tmp_path):
    """ColmapDataset over a scene folder (no ray cache): every llff_hold_step-th view is validation, the rest training
    (datasets.py:330-341); a sample of the validation split is one whole image with its own bounds."""
    import numpy as np
    from nerfmeshes_amd import synthetic as S
    from nerfmeshes_amd.data import ColmapDataset, DatasetType
    from nerfmeshes_amd.nerf import CfgNode
    gold = _llff_scene(tmp_path)
    hp = S.hparams(dataset_type="colmap", near=0.0, far=1.0)
    hp.update({"dataset.basedir": str(tmp_path), "dataset.llff_downsample_factor": 4, "dataset.llff_hold_step": 4,
               "dataset.caching.use_caching": False})
    cfg = CfgNode(nest_dict(hp, sep="."))

    def reader(split):          # the constructor would go on to generate every view's rays on the GPU: the file reader alone here
        ds = ColmapDataset.__new__(ColmapDataset)
        ds.cfg, ds.type, ds.downscale_factor, ds.spherify, ds.path = cfg, split, 4, False, str(tmp_path / "cache")
        return ds.load_dataset()

    val, train = reader(DatasetType.VALIDATION), reader(DatasetType.TRAIN)
    assert val.size == 3 and train.size == 6 and val.hwf == (6, 8, 30)
    np.testing.assert_allclose(val.poses.numpy(), gold["forward_poses"][[0, 4, 8], :3, :4], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(train.ray_bounds.numpy(), gold["forward_bounds"][[1, 2, 3, 5, 6, 7]], rtol=2e-6)
    assert tuple(val.ray_targets.shape) == (3, 6, 8, 3)
