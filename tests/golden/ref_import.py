"""Import the UNMODIFIED reference (`/root/reference/src`) in this container.

Only used to (a) generate the committed golden fixtures and (b) validate the
oracle restatement on the build container.  /root/reference does not exist on
the GPU box: nothing in `-m gpu` tests, smoke() or bench.py imports this.

The reference depends on packages that are not installed offline
(torchvision, pytorch_lightning, pytorch3d, skimage, cv2, imageio, OpenEXR,
tensorboard).  None of them is on the arithmetic path we pin, so they are
replaced with inert stub modules.
"""
import collections
import collections.abc
import importlib
import os
import sys
import types

REF_SRC = os.environ.get("NERFMESHES_REFERENCE", "/root/reference") + "/src"


def available():
    return os.path.isdir(REF_SRC)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch

    if not hasattr(collections, "MutableMapping"):
        collections.MutableMapping = collections.abc.MutableMapping  # model_helpers.py:10

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

        def __getattr__(self, k):
            return _Any()

    class LightningModule(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", ToPILImage=_Any)
    tv.utils = _stub("torchvision.utils")
    pl = _stub("pytorch_lightning", LightningModule=LightningModule, Callback=object,
               Trainer=_Any, seed_everything=lambda *a, **k: None)
    pl.callbacks = _stub("pytorch_lightning.callbacks", Callback=object, ModelCheckpoint=_Any)
    tbl = type("TensorBoardLogger", (), {"NAME_HPARAMS_FILE": "hparams.yaml", "__init__": lambda s, *a, **k: None})
    pl.loggers = _stub("pytorch_lightning.loggers", TensorBoardLogger=tbl)
    pl.profiler = _stub("pytorch_lightning.profiler", AdvancedProfiler=_Any)
    p3 = _stub("pytorch3d")
    p3.ops = _stub("pytorch3d.ops", sample_points_from_meshes=_Any())
    p3.loss = _stub("pytorch3d.loss", chamfer_distance=_Any())
    p3.structures = _stub("pytorch3d.structures", Meshes=_Any)
    _stub("torch.utils.tensorboard", SummaryWriter=_Any)
    sk = _stub("skimage")
    sk.measure = _stub("skimage.measure", marching_cubes=_Any())
    sk.transform = _stub("skimage.transform")
    _stub("cv2")
    _stub("imageio")
    _stub("OpenEXR")
    _stub("Imath")
    mpl = sys.modules.get("matplotlib")
    if mpl is None:
        try:
            importlib.import_module("matplotlib")
        except Exception:
            m = _stub("matplotlib")
            m.pyplot = _stub("matplotlib.pyplot")
            _stub("mpl_toolkits")
            _stub("mpl_toolkits.mplot3d", Axes3D=_Any)


_loaded = {}


def load():
    """Returns (nerf, models) reference packages."""
    if _loaded:
        return _loaded["nerf"], _loaded["models"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_SRC)
    install_stubs()
    # the reference packages are called `nerf`, `models`, `data` -- they must be
    # imported under those names (absolute intra-package imports).
    sys.path.insert(0, REF_SRC)
    try:
        nerf = importlib.import_module("nerf")
        models = importlib.import_module("models")
    finally:
        sys.path.remove(REF_SRC)
    _loaded["nerf"], _loaded["models"] = nerf, models
    return nerf, models
