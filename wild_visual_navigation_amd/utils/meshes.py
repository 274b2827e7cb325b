"""Footprint geometry helpers -- wild_visual_navigation/utils/meshes.py:77-165 (``make_plane``, ``make_dense_plane``,
``make_polygon_from_points``) and kornia's ``transform_points`` restated on torch tensors (kornia is not a dependency).
A few dozen points per robot-state callback: host-level plumbing, not kernel work; the pixel work that consumes them is
csrc/supervision.hip."""
import torch


def transform_points(trans_01: torch.Tensor, points_1: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.linalg.transform_points: [B,D+1,D+1] x [B,N,D] -> [B,N,D] (homogeneous multiply, then the
    convert_points_from_homogeneous division with its eps = 1e-8 rule)."""
    ones = torch.ones_like(points_1[..., :1])
    ph = torch.cat([points_1, ones], dim=-1)
    out = torch.matmul(ph, trans_01.transpose(-1, -2))
    z = out[..., -1:]
    scale = torch.where(z.abs() > 1e-8, 1.0 / (z + 1e-8), torch.ones_like(z))
    return scale * out[..., :-1]


def make_plane(x=None, y=None, z=None, pose=torch.eye(4), grid_size=10):
    """meshes.py:77-124: rectangle outline in the plane whose coordinate is None, refined along its edges, de-duplicated with
    torch.unique(dim=0) (which also SORTS the rows), transformed by ``pose``."""
    if x is None:
        points = torch.tensor([[0.0, y / 2, z / 2], [0.0, -y / 2, z / 2], [0.0, -y / 2, -z / 2], [0.0, y / 2, -z / 2]])
    elif y is None:
        points = torch.tensor([[x / 2, 0.0, z / 2], [x / 2, 0.0, -z / 2], [-x / 2, 0.0, -z / 2], [-x / 2, 0.0, z / 2]])
    elif z is None:
        points = torch.tensor([[x / 2, y / 2, 0.0], [x / 2, -y / 2, 0.0], [-x / 2, -y / 2, 0.0], [-x / 2, y / 2, 0.0]])
    else:
        raise TypeError("make_plane requires just 2 inputs to be set")
    points = points.float().to(pose.device)
    finer = [points]
    if grid_size > 0:
        w_steps = torch.linspace(0, 1, steps=grid_size).to(pose.device)
        for i in range(4):
            for w in w_steps:
                finer.append(torch.lerp(points[i], points[(i + 1) % 4], w)[None])
    finer = torch.unique(torch.cat(finer), dim=0)
    if pose.dim() == 2:
        pose = pose[None]
    return transform_points(pose, finer[None])[0]


def make_dense_plane(x=None, y=None, z=None, pose=torch.eye(4), grid_size=5):
    """meshes.py:127-151."""
    dev = pose.device
    if x is None:
        xs, ys, zs = torch.linspace(0.0, 0.0, grid_size), torch.linspace(-y / 2, y / 2, grid_size), torch.linspace(-z / 2, z / 2, grid_size)
    elif y is None:
        xs, ys, zs = torch.linspace(-x / 2, x / 2, grid_size), torch.linspace(0.0, 0.0, grid_size), torch.linspace(-z / 2, z / 2, grid_size)
    elif z is None:
        xs, ys, zs = torch.linspace(-x / 2, x / 2, grid_size), torch.linspace(-y / 2, y / 2, grid_size), torch.linspace(0.0, 0.0, grid_size)
    else:
        raise TypeError("make_plane requires just 2 inputs to be set")
    gx, gy, gz = torch.meshgrid(xs.to(dev), ys.to(dev), zs.to(dev), indexing="xy")
    points = torch.cat((gx.ravel()[:, None], gy.ravel()[:, None], gz.ravel()[:, None]), dim=1)
    if pose.dim() == 2:
        pose = pose[None]
    return transform_points(pose, points[None])[0]


def make_polygon_from_points(points: torch.Tensor, grid_size=10):
    """meshes.py:154-163: every edge of the (sorted) polygon refined into ``grid_size`` lerp points."""
    B = points.shape[0]
    w_steps = torch.linspace(0, 1, steps=grid_size).to(points.device)
    finer = []
    for i in range(B):
        for w in w_steps:
            finer.append(torch.lerp(points[i], points[(i + 1) % B], w)[None])
    return torch.cat(finer, dim=0)
