"""`install()` registers this package under the module names the reference's scripts import
(`models`, `nerf`, `nerf.tree`, `data`, `lightning_modules`, `mesh_nerf`), so that `eval_nerf.py`-style code and pickled
BuFF checkpoints (`nerf.tree.Node`) resolve to the MI355X implementation (route A)."""
import sys


def install():
    from . import data, lightning_modules, mesh_nerf, models, nerf
    from .models import model_base, model_buff, model_helpers, model_nerf
    sys.modules.update({
        "models": models, "models.model_base": model_base, "models.model_nerf": model_nerf,
        "models.model_buff": model_buff, "models.model_helpers": model_helpers,
        "nerf": nerf, "nerf.tree": nerf.tree, "nerf.nerf_helpers": nerf.nerf_helpers, "nerf.modules": nerf.modules,
        "nerf.models": nerf.models, "nerf.cfgnode": nerf.cfgnode,
        "data": data, "data.data_helpers": data.data_helpers, "data.datasets": data.datasets,
        "lightning_modules": lightning_modules, "mesh_nerf": mesh_nerf,
    })
    return models, nerf
