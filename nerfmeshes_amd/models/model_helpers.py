"""hparams <-> cfg helpers (mirror of /root/reference/src/models/model_helpers.py:6-35)."""
from collections.abc import MutableMapping


def flatten_dict(d, parent_key="", sep="_"):
    """Nested mapping -> flat {"a.b.c": v}; the layout Lightning writes to hparams.yaml."""
    flat = {}
    for k, v in d.items():
        key = f"{parent_key}{sep}{k}" if parent_key else k
        if isinstance(v, MutableMapping):
            flat.update(flatten_dict(v, key, sep=sep))
        else:
            flat[key] = v
    return flat


def nest_dict(flat, sep="_"):
    """Inverse of flatten_dict; idempotent on keys that contain no separator."""
    out = {}
    for key, v in flat.items():
        node = out
        parts = key.split(sep)
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return out


def intervals_to_ray_points(point_intervals, ray_directions, ray_origin):
    """p = o + d * t.  On the hot path this is fused into the MLP kernel's prologue (nm_mlp_eval_rays);
    the standalone form is kept for callers that want the points themselves (torch op on any device)."""
    return ray_origin[..., None, :] + ray_directions[..., None, :] * point_intervals[..., :, None]
