"""The two liegroups routines the path uses (liegroups is not a dependency): the translational part of the SE(3) logarithm
(``BaseNode.distance_to``, nodes.py:73-91) and ``SO3.from_rpy`` (``SupervisionNode.get_untraversable_plane``, nodes.py:512-548).
Scalar 3x3 / 4x4 math per graph operation: host plumbing."""
import math

import torch


def _normalize_rotation(R: torch.Tensor) -> torch.Tensor:
    """liegroups SO3.normalize: nearest rotation by SVD (det forced to +1)."""
    U, _, Vt = torch.linalg.svd(R)
    S = torch.eye(3, dtype=R.dtype, device=R.device)
    S[2, 2] = torch.linalg.det(U) * torch.linalg.det(Vt)
    return U @ S @ Vt


def so3_log(R: torch.Tensor) -> torch.Tensor:
    cos = (0.5 * torch.trace(R) - 0.5).clamp(-1.0, 1.0)
    angle = torch.acos(cos)
    if float(angle) < 1e-7:
        W = R - torch.eye(3, dtype=R.dtype, device=R.device)
    else:
        W = (0.5 * angle / torch.sin(angle)) * (R - R.T)
    return torch.stack([W[2, 1], W[0, 2], W[1, 0]])


def _wedge(p: torch.Tensor) -> torch.Tensor:
    z = torch.zeros((), dtype=p.dtype, device=p.device)
    return torch.stack([torch.stack([z, -p[2], p[1]]), torch.stack([p[2], z, -p[0]]), torch.stack([-p[1], p[0], z])])


def so3_inv_left_jacobian(phi: torch.Tensor) -> torch.Tensor:
    angle = phi.norm()
    eye = torch.eye(3, dtype=phi.dtype, device=phi.device)
    if float(angle) < 1e-7:
        return eye - 0.5 * _wedge(phi)
    axis = phi / angle
    ha = 0.5 * angle
    cot = 1.0 / torch.tan(ha)
    return ha * cot * eye + (1 - ha * cot) * torch.outer(axis, axis) - ha * _wedge(axis)


def se3_log_translation_norm(T: torch.Tensor) -> torch.Tensor:
    """|| SE3.from_matrix(T, normalize=True).log()[:3] ||: rho = J^-1(phi) t."""
    R = _normalize_rotation(T[:3, :3])
    phi = so3_log(R)
    rho = so3_inv_left_jacobian(phi) @ T[:3, 3]
    return rho.norm()


def so3_from_rpy(roll: float, pitch: float, yaw: float) -> torch.Tensor:
    """liegroups SO3.from_rpy: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    Rx = torch.tensor([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=torch.float32)
    Ry = torch.tensor([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=torch.float32)
    Rz = torch.tensor([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=torch.float32)
    return Rz @ Ry @ Rx
