"""LLFF / COLMAP forward-facing scene reader -- what `ColmapDataset.load_dataset` consumes
(/root/reference/src/data/datasets.py:325-357 calls /root/reference/src/data/loaders/load_llff.py:278-354 `load_llff_data`):

    <basedir>/poses_bounds.npy        (N, 17): a 3 x 5 block [R | t | (H, W, f)] in LLFF's (down, right, back) axis order + the
                                      near / far depth of the view
    <basedir>/images[_<factor>]/*     one JPG / PNG per pose, sorted by name (the down-scaled folder must exist: the reference
                                      shells out to ImageMagick's `mogrify` to make it; here a missing folder is an error that
                                      says so)

-> `images (N, H, W, 3)` in [0, 1], `poses (N, 3, 5)` camera-to-world [R | t | hwf] in the (right, up, back) convention of
`get_ray_bundle`, `bounds (N, 2)`, `render_poses (120, 3, 5)` (a spiral through the average pose, or a circle for spherified
scenes), `i_test` = the view closest to the average pose -- the arithmetic in fp64 / fp32 exactly where the reference has it, so
that the arrays agree with its own to the last bits (tests/test_data_feed.py holds them against arrays the unmodified reference
produced from the same folder).  Host-side numpy; nothing here is on the GPU path.  Images are decoded with Pillow.
"""
import os

import numpy as np

_IMAGE_SUFFIXES = ("JPG", "jpg", "png")


def _unit(v):
    return v / np.linalg.norm(v)


def _image_files(folder):
    return [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith(_IMAGE_SUFFIXES)]


def _decode(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def read_scene(basedir, factor=None, load_images=True):
    """poses_bounds.npy + the image folder -> (poses (3, 5, N) fp64 in the file's axis order with hwf rescaled to the images that
    are actually read, bounds (2, N), images (H, W, 3, N) fp64 in [0, 1])."""
    table = np.load(os.path.join(basedir, "poses_bounds.npy"))
    poses = table[:, :-2].reshape(-1, 3, 5).transpose(1, 2, 0)
    bounds = table[:, -2:].transpose(1, 0)
    # factor None: <basedir>/images as they are; otherwise <basedir>/images_<factor> (for factor 1 the originals serve when that
    # folder was never made: the reference's own would be their PNG copies)
    folder = os.path.join(basedir, "images" if factor is None else f"images_{factor}")
    if factor == 1 and not os.path.isdir(folder):
        folder = os.path.join(basedir, "images")
    if factor is None:
        factor = 1
    if not os.path.isdir(folder):
        raise FileNotFoundError(
            f"{folder}: the images down-scaled by dataset.llff_downsample_factor = {factor} are not there.  The reference creates "
            f"them with ImageMagick (`mogrify -resize {100.0 / factor}% -format png`) on first use; do that once, or set the factor "
            "to 1 to read <basedir>/images.")
    files = _image_files(folder)
    if poses.shape[-1] != len(files):
        raise ValueError(f"{basedir}: {len(files)} images in {os.path.basename(folder)} but {poses.shape[-1]} poses in poses_bounds.npy")
    height, width = _decode(files[0]).shape[:2]
    poses[0, 4, :], poses[1, 4, :] = height, width
    poses[2, 4, :] = poses[2, 4, :] * 1.0 / factor           # the focal length follows the down-scaling
    if not load_images:
        return poses, bounds, None
    images = np.stack([_decode(f)[..., :3] / 255.0 for f in files], -1)
    return poses, bounds, images


def look_along(z, up, position):
    """Camera-to-world (3, 4) looking along -z with `up` roughly up."""
    back = _unit(z)
    right = _unit(np.cross(up, back))
    return np.stack([right, _unit(np.cross(back, right)), back, position], 1)


def average_pose(poses):
    """(N, 3, 5) -> the (3, 5) pose at the mean position looking along the summed view axes (hwf of view 0)."""
    return np.concatenate([look_along(_unit(poses[:, :3, 2].sum(0)), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0)), poses[0, :3, -1:]], 1)


def _homogeneous(p):
    """(..., 3, 4) -> (..., 4, 4)."""
    row = np.broadcast_to(np.array([0, 0, 0, 1.0]), p.shape[:-2] + (1, 4))
    return np.concatenate([p, row], -2)


def recenter(poses):
    """Express every pose in the frame of the average pose."""
    out = poses + 0
    world_from_avg = _homogeneous(average_pose(poses)[:3, :4])
    out[:, :3, :4] = (np.linalg.inv(world_from_avg) @ _homogeneous(poses[:, :3, :4]))[:, :3, :4]
    return out


def spiral_path(c2w, up, radii, focal, zrate, rotations, views):
    """`views` poses on a spiral around c2w (3, 5), all looking at the point `focal` in front of it."""
    radii = np.array(list(radii) + [1.0])
    out = []
    for theta in np.linspace(0.0, 2.0 * np.pi * rotations, views + 1)[:-1]:
        position = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.0]) * radii)
        target = np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.0]))
        out.append(np.concatenate([look_along(_unit(position - target), up, position), c2w[:, 4:5]], 1))
    return out


def spherify(poses, bounds):
    """Inward-facing captures: move the point closest to all optical axes to the origin, scale the cameras onto the unit
    sphere, and return a circle of 120 render poses at the cameras' mean height.  `bounds` is scaled IN PLACE, as the
    reference does."""
    axes, origins = poses[:, :3, 2:3], poses[:, :3, 3:4]
    # the point of least summed squared distance to the optical axes
    projector = np.eye(3) - axes * np.transpose(axes, [0, 2, 1])
    rhs = -projector @ origins
    centre = np.squeeze(-np.linalg.inv((np.transpose(projector, [0, 2, 1]) @ projector).mean(0)) @ rhs.mean(0))
    up = _unit((poses[:, :3, 3] - centre).mean(0))
    side = _unit(np.cross([0.1, 0.2, 0.3], up))
    frame = np.stack([side, _unit(np.cross(up, side)), up, centre], 1)
    local = np.linalg.inv(_homogeneous(frame[None])) @ _homogeneous(poses[:, :3, :4])
    radius = np.sqrt(np.mean(np.sum(np.square(local[:, :3, 3]), -1)))
    scale = 1.0 / radius
    local[:, :3, 3] *= scale
    bounds *= scale
    radius *= scale
    height = np.mean(local[:, :3, 3], 0)[2]
    ring = np.sqrt(radius ** 2 - height ** 2)
    circle = []
    for theta in np.linspace(0.0, 2.0 * np.pi, 120):
        position = np.array([ring * np.cos(theta), ring * np.sin(theta), height])
        back = _unit(position)
        right = _unit(np.cross(back, np.array([0, 0, -1.0])))
        circle.append(np.stack([right, _unit(np.cross(back, right)), back, position], 1))
    circle = np.stack(circle, 0)
    hwf = poses[0, :3, -1:]
    circle = np.concatenate([circle, np.broadcast_to(hwf, circle[:, :3, -1:].shape)], -1)
    local = np.concatenate([local[:, :3, :4], np.broadcast_to(hwf, local[:, :3, -1:].shape)], -1)
    return local, circle, bounds


def load_llff_data(basedir, factor=8, recenter_poses=True, bd_factor=0.75, spherify_poses=False, path_zflat=False):
    """The reference's entry point (load_llff.py:278-354): -> images, poses, bounds, render_poses, i_test."""
    poses, bounds, images = read_scene(basedir, factor=factor)
    print("Loaded", basedir, bounds.min(), bounds.max())
    # LLFF stores the rotation columns as (down, right, back): -> (right, up, back); views to axis 0
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(images, -1, 0).astype(np.float32)
    bounds = np.moveaxis(bounds, -1, 0).astype(np.float32)
    scale = 1.0 if bd_factor is None else 1.0 / (bounds.min() * bd_factor)       # nearest depth -> 1 / bd_factor
    poses[:, :3, 3] *= scale
    bounds *= scale
    if recenter_poses:
        poses = recenter(poses)
    if spherify_poses:
        poses, render_poses, bounds = spherify(poses, bounds)
    else:
        path_pose = average_pose(poses)
        up = _unit(poses[:, :3, 1].sum(0))
        near, far = bounds.min() * 0.9, bounds.max() * 5.0
        blend = 0.75
        focal = 1.0 / ((1.0 - blend) / near + blend / far)                       # a depth between the two, in disparity
        radii = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        rotations, views = 2, 120
        if path_zflat:
            path_pose[:3, 3] = path_pose[:3, 3] + (-near * 0.1) * path_pose[:3, 2]
            radii[2] = 0.0
            rotations, views = 1, 60
        render_poses = spiral_path(path_pose, up, radii, focal, zrate=0.5, rotations=rotations, views=views)
    render_poses = np.array(render_poses).astype(np.float32)
    centre = average_pose(poses)
    i_test = int(np.argmin(np.sum(np.square(centre[:3, 3] - poses[:, :3, 3]), -1)))
    print("HOLDOUT view is", i_test)
    return images.astype(np.float32), poses.astype(np.float32), bounds, render_poses, i_test
