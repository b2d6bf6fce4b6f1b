"""Helper functions the three scripts import by name (mirror of /root/reference/src/nerf/nerf_helpers.py)."""
import os

import numpy as np
import torch

from .. import hip_ops


def img2mse(img_src, img_tgt):
    return torch.nn.functional.mse_loss(img_src, img_tgt)


def mse2psnr(mse):
    """nerf_helpers.py:17-23: PSNR = -10 log10(mse), MAX = 1."""
    mse = torch.as_tensor(mse)
    if mse == 0:
        mse = torch.full_like(mse, 1e-5, dtype=torch.float32)
    return -10.0 * torch.log10(mse)


# colours of the depth-residual point clouds (nerf_helpers.py:8-11)
POINT_GROUND_TRUTH = torch.tensor([0., 0., 255.])
POINT_OUT_TRUE = torch.tensor([0., 255., 0.])
POINT_OUT_FALSE_VOID = torch.tensor([0., 0., 0.])
POINT_OUT_FALSE_SURFACE = torch.tensor([255., 0., 0.])


def create_point_cloud(ray_origins, ray_directions, depth, color, mask=None):
    """nerf_helpers.py:56-64: points o + d * depth (of the masked rays), one constant colour, normals = -d."""
    if mask is not None:
        ray_directions, depth = ray_directions[mask], depth[mask]
    vertices = (ray_origins + ray_directions * depth[..., None]).view(-1, 3)
    return vertices, color.to(vertices.device).expand(vertices.shape), -ray_directions.view(-1, 3)


def get_point_clouds(ray_origins, ray_directions, depth_output, depth_target=None, threshold=0.2, empty=0.):
    """nerf_helpers.py:26-53: the rendered depths as a coloured point cloud; with target depths, four clouds
    (target / hits within `threshold` / misses over empty space / misses over the surface) concatenated per field."""
    if depth_target is None:
        return create_point_cloud(ray_origins, ray_directions, depth_output, POINT_GROUND_TRUTH)
    hit = torch.abs(depth_output - depth_target) < threshold
    clouds = [create_point_cloud(ray_origins, ray_directions, depth_target, POINT_GROUND_TRUTH),
              create_point_cloud(ray_origins, ray_directions, depth_output, POINT_OUT_TRUE, hit),
              create_point_cloud(ray_origins, ray_directions, depth_output, POINT_OUT_FALSE_VOID, (depth_target == empty) & ~hit),
              create_point_cloud(ray_origins, ray_directions, depth_output, POINT_OUT_FALSE_SURFACE, (depth_target != empty) & ~hit)]
    return [torch.cat(field, dim=0) for field in zip(*clouds)]


def comp_depth(depth_output, depth_target, empty_value=0.):
    """nerf_helpers.py:67-83: depth MSE overall / over empty pixels / over surface pixels, and the mean signed error."""
    mse = torch.nn.functional.mse_loss
    surface = depth_target > empty_value
    return (mse(depth_output, depth_target), mse(depth_output[~surface], depth_target[~surface]),
            mse(depth_output[surface], depth_target[surface]), (depth_output[surface] - depth_target[surface]).mean())


def batchify(*data, batch_size=1024, device="cpu", progress=True):
    """nerf_helpers.py:114-139: yield aligned slices of every tensor, moved to `device`."""
    size = data[0].shape[0]
    assert all(s is None or s.shape[0] == size for s in data), "Sizes of tensors must match for dimension 0."

    def gen():
        for start in range(0, size, batch_size):
            yield [s[start:start + batch_size].to(device) if s is not None else s for s in data]

    if not progress:
        return gen()
    try:
        from tqdm import tqdm
        return tqdm(gen(), total=(size - 1) // batch_size + 1)
    except ImportError:
        return gen()


def meshgrid_xy(tensor1, tensor2):
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


def cumprod_exclusive(tensor):
    """nerf_helpers.py:199-223 (kept for API compatibility; the render path does the scan in-kernel)."""
    c = torch.roll(torch.cumprod(tensor, -1), 1, -1)
    c[..., 0] = 1.0
    return c


def get_ray_bundle(height, width, focal_length, tform_cam2world):
    """nerf_helpers.py:226-277 on the GPU: (origin (3,), directions (H, W, 3))."""
    dev = tform_cam2world.device if isinstance(tform_cam2world, torch.Tensor) and tform_cam2world.is_cuda else "cuda"
    origin, dirs = hip_ops.ray_bundle(tform_cam2world, int(height), int(width), float(focal_length), device=dev)
    return origin, dirs.reshape(int(height), int(width), 3)


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf_helpers.py:280-307 -> nm_ndc_rays (one HIP kernel, op for op as the reference's torch expressions)."""
    return hip_ops.ndc_rays(H, W, focal, near, rays_o, rays_d)


def cast_to_image(tensor):
    """(H, W, 3) float -> (3, H, W) uint8 numpy (nerf_helpers.py:155-181 family)."""
    img = (tensor.detach().cpu().clamp(0.0, 1.0) * 255).to(torch.uint8).numpy()
    return np.moveaxis(img, [-1], [0])


def cast_to_pil_image(tensor):
    return (tensor.detach().cpu().clamp(0.0, 1.0) * 255).to(torch.uint8).numpy()


def cast_to_disparity_image(tensor, white_background=False):
    """(H, W) -> uint8 image, min-max normalised; exact zeros become white if requested."""
    img = (tensor - tensor.min()) / (tensor.max() - tensor.min())
    img = (img.clamp(0.0, 1.0) * 255).byte()
    if white_background:
        img[img == 0] = 255
    return img.detach().cpu().numpy()


def export_point_cloud(it, ray_origins, ray_directions, depth_fine, dep_target):
    """nerf_helpers.py:142-152: rendered (red) and target (blue) depth points of one image -> `<it:04d>.obj`."""
    rendered = (ray_origins + ray_directions * depth_fine[..., None]).view(-1, 3)
    target = (ray_origins + ray_directions * dep_target[..., None]).view(-1, 3)
    red, blue = torch.zeros_like(rendered), torch.zeros_like(target)
    red[:, 0], blue[:, 2] = 1.0, 1.0
    back = -ray_directions.view(-1, 3)
    export_obj(torch.cat((rendered, target), 0), [], torch.cat((red, blue), 0), torch.cat((back, back), 0), f"{it:04d}.obj")


def export_obj(vertices, triangles, diffuse, normals, filename):
    """nerf_helpers.py:86-111 text format: `v x y z [r g b]`, `vn x y z`, `f a//a b//b c//c` (1-based), numbers exactly
    as the reference's `"{}".format(tensor_element)` prints them (repr of the fp32 value widened to a Python float).
    Formatting runs in the native library on all host threads (nm_export_obj): the per-element Python writer needs
    10.6 s for the 480^3 mesh (2.3 M lines, 209 MB) -- 14x the whole GPU extraction."""
    import ctypes as C
    from .. import _lib

    def host(x, dtype):
        if isinstance(x, torch.Tensor):
            x = x.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(x), dtype=dtype).reshape(-1, 3)

    v, c, n, t = host(vertices, np.float32), host(diffuse, np.float32), host(normals, np.float32), host(triangles, np.int32)
    print("Writing to obj...")
    ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(_lib.load().nm_export_obj(ptr(v), len(v), ptr(c), len(c), ptr(n), len(n), ptr(t), len(t),
                                         os.fsencode(filename)), "nm_export_obj")
    print(f"Finished writing to {filename} with {len(v)} vertices")
