"""`install()` registers this package under the module names the reference's scripts import (`models`, `nerf`,
`nerf.tree`, `data`, `data.datasets`, `lightning_modules`, `mesh_nerf`), so that the UNMODIFIED
`/root/reference/src/{eval_nerf,mesh_nerf,train_nerf}.py` and pickled BuFF checkpoints (`nerf.tree.Node`) resolve to the
MI355X implementation (route A; `tests/test_reference_scripts.py` executes the three scripts this way).

The scripts also import third-party packages that an offline MI355X image may lack.  `install()` fills in ONLY the ones
whose import fails, with the narrow slice the scripts use:

| import in the reference | stand-in (only if the real package is missing) |
|---|---|
| `pytorch_lightning` (`Trainer`, `seed_everything`, `.callbacks.ModelCheckpoint/Callback`, `.loggers.TensorBoardLogger`, `.core.memory.ModelSummary`, `.profiler.AdvancedProfiler`) | `nerfmeshes_amd.lightning_compat` |
| `skimage.measure.marching_cubes` (`mesh_nerf.py:79`) | `nm_mc_count/nm_mc_emit` on the GPU, numpy in / numpy out (bitwise what scikit-image returns) |
| `imageio.imwrite / imread` (`eval_nerf.py:88-101`) | Pillow |
| `pytorch3d.structures.Meshes`, `pytorch3d.ops`, `pytorch3d.loss` (`mesh_nerf.py:8`; only the dead chamfer branch calls them) | placeholders that raise when called |
"""
import importlib
import sys
import types


def _missing(name):
    if name in sys.modules:
        return False
    try:
        importlib.import_module(name)
        return False
    except Exception:  # noqa: BLE001  (ImportError, or a broken optional dependency)
        return True


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__nerfmeshes_amd_stand_in__ = True
    sys.modules[name] = m
    return m


def _unavailable(what):
    def fn(*a, **k):
        raise RuntimeError(f"{what} is not installed and nerfmeshes_amd provides no replacement for it")
    return fn


def marching_cubes(volume, level=None, **kwargs):
    """`skimage.measure.marching_cubes(volume, level)` with scikit-image's defaults (Lewiner, spacing 1, descent,
    step 1, allow_degenerate) on the GPU: numpy (n0,n1,n2) in, numpy (verts, faces, normals, values) out."""
    import numpy as np
    import torch
    from . import hip_ops
    unsupported = {k: v for k, v in kwargs.items() if v is not None and (k, v) not in (
        ("spacing", (1.0, 1.0, 1.0)), ("gradient_direction", "descent"), ("step_size", 1), ("allow_degenerate", True),
        ("method", "lewiner"), ("use_classic", False))}
    if unsupported:
        raise NotImplementedError(f"marching_cubes: only scikit-image's defaults are implemented, got {unsupported}")
    vol = torch.as_tensor(np.ascontiguousarray(volume, dtype=np.float32)).cuda()
    if level is None:
        level = 0.5 * (float(vol.min()) + float(vol.max()))
    return tuple(t.cpu().numpy() for t in hip_ops.marching_cubes(vol, float(level)))


def _imwrite(path, image, **kwargs):
    import numpy as np
    from PIL import Image
    Image.fromarray(np.asarray(image)).save(str(path))


def _imread(path, **kwargs):
    import numpy as np
    from PIL import Image
    with Image.open(str(path)) as im:
        return np.asarray(im)


def install_third_party():
    """Register stand-ins for the third-party imports of the three scripts that are missing here; returns their names."""
    from . import lightning_compat as lc
    filled = []
    if _missing("pytorch_lightning"):
        pl = _module("pytorch_lightning", Trainer=lc.Trainer, seed_everything=lc.seed_everything, Callback=lc.Callback,
                     LightningModule=lc.LightningModule, __version__=lc.__version__, __path__=[])
        pl.callbacks = _module("pytorch_lightning.callbacks", Callback=lc.Callback, ModelCheckpoint=lc.ModelCheckpoint)
        pl.loggers = _module("pytorch_lightning.loggers", TensorBoardLogger=lc.TensorBoardLogger)
        pl.core = _module("pytorch_lightning.core", LightningModule=lc.LightningModule, __path__=[])
        pl.core.memory = _module("pytorch_lightning.core.memory", ModelSummary=lc.ModelSummary)
        pl.profiler = _module("pytorch_lightning.profiler", AdvancedProfiler=lc.AdvancedProfiler)
        filled.append("pytorch_lightning")
    if _missing("skimage"):
        sk = _module("skimage", __path__=[])
        sk.measure = _module("skimage.measure", marching_cubes=marching_cubes, marching_cubes_lewiner=marching_cubes)
        filled.append("skimage")
    if _missing("imageio"):
        _module("imageio", imwrite=_imwrite, imsave=_imwrite, imread=_imread)
        filled.append("imageio")
    if _missing("pytorch3d"):
        p3 = _module("pytorch3d", __path__=[])
        p3.structures = _module("pytorch3d.structures", Meshes=_unavailable("pytorch3d.structures.Meshes"))
        p3.ops = _module("pytorch3d.ops", sample_points_from_meshes=_unavailable("pytorch3d.ops.sample_points_from_meshes"))
        p3.loss = _module("pytorch3d.loss", chamfer_distance=_unavailable("pytorch3d.loss.chamfer_distance"))
        filled.append("pytorch3d")
    return filled


def install(third_party=True):
    """Alias the reference's module names to this package; with `third_party` also fill in missing third-party imports
    (see the table above).  Returns (models, nerf)."""
    if third_party:
        install_third_party()
    from . import data, lightning_modules, mesh_nerf, models, nerf
    from .data import loaders
    from .data.loaders import load_blender
    from .models import model_base, model_buff, model_helpers, model_nerf
    sys.modules.update({
        "models": models, "models.model_base": model_base, "models.model_nerf": model_nerf,
        "models.model_buff": model_buff, "models.model_helpers": model_helpers,
        "nerf": nerf, "nerf.tree": nerf.tree, "nerf.nerf_helpers": nerf.nerf_helpers, "nerf.modules": nerf.modules,
        "nerf.models": nerf.models, "nerf.cfgnode": nerf.cfgnode,
        "data": data, "data.data_helpers": data.data_helpers, "data.datasets": data.datasets,
        "data.loaders": loaders, "data.loaders.load_blender": load_blender,
        "lightning_modules": lightning_modules, "mesh_nerf": mesh_nerf,
    })
    return models, nerf
