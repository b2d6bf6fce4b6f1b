"""Attribute-access configuration tree (mirror of the reference's yacs-style `nerf.CfgNode`,
/root/reference/src/nerf/cfgnode.py:36-141 -- only the behaviour the hot path and the three scripts
rely on: nested dict -> nested CfgNode, `cfg.a.b` reads/writes, `hasattr`, `**cfg.models.coarse`)."""
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v, (key_list or []) + [k]) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        return yaml.safe_dump(self.to_dict(), **kwargs)

    def clone(self):
        return copy.deepcopy(self)

    @classmethod
    def load_cfg(cls, file_or_str):
        data = yaml.safe_load(file_or_str.read() if hasattr(file_or_str, "read") else file_or_str)
        return cls(data)
