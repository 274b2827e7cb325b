"""Oracle: supervision-mask path.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy fp32 with one rounding per operation and a fixed operation order,
  wild_visual_navigation/image_projector/image_projector.py:126-197  (ImageProjector.project / project_and_render)
  wild_visual_navigation/traversability_estimator/traversability_estimator.py:261-289 (render, * traversability, torch.fmin)
whose arithmetic lives in kornia (PinholeCamera.project, transform_points, convert_points_from_homogeneous,
utils.draw.draw_convex_polygon / _get_convex_edges).  kornia is absent from this image and from /root/reference (setup.py
asks for kornia>=0.6.5, un-vendored): PARITY UNPINNED for those routines -- they are restated from the published kornia
(>= 0.6.7) source; ``tests/test_oracle_supervision.py`` cross-checks the fill against an independent half-plane test.
One documented deviation: T_cw is the rigid inverse [R^T | -R^T t] instead of torch.inverse (equal to ~1e-7 for SE(3) poses).
"""
import numpy as np

f32 = np.float32


def project_points_raw(K: np.ndarray, pose_cam_in_world: np.ndarray, pts: np.ndarray):
    """ImageProjector.project (image_projector.py:126-150): K [4,4], pose [4,4], pts [N,3] -> (raw pixel coordinates [N,2],
    finite also behind the camera; camera-frame depth z [N]; valid_z = z >= 0)."""
    K, T, P = K.astype(f32), pose_cam_in_world.astype(f32), pts.astype(f32)
    X, Y, Z = P[:, 0], P[:, 1], P[:, 2]
    pc = []
    for r in range(3):
        r0, r1, r2 = T[0, r], T[1, r], T[2, r]
        tc = -((r0 * T[0, 3] + r1 * T[1, 3]) + r2 * T[2, 3])
        pc.append(((r0 * X + r1 * Y) + r2 * Z) + tc)
    xp = ((K[0, 0] * pc[0] + K[0, 1] * pc[1]) + K[0, 2] * pc[2]) + K[0, 3]
    yp = ((K[1, 0] * pc[0] + K[1, 1] * pc[1]) + K[1, 2] * pc[2]) + K[1, 3]
    zp = ((K[2, 0] * pc[0] + K[2, 1] * pc[1]) + K[2, 2] * pc[2]) + K[2, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(np.abs(zp) > f32(1e-8), f32(1.0) / (zp + f32(1e-8)), f32(1.0)).astype(f32)
    u, v = (xp * s).astype(f32), (yp * s).astype(f32)
    return np.stack([u, v], axis=1), pc[2].astype(f32)


def project_points(K: np.ndarray, pose_cam_in_world: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """The polygon vertices as project_and_render draws them: behind the camera -> NaN (image_projector.py:180)."""
    uv, z = project_points_raw(K, pose_cam_in_world, pts)
    uv = uv.copy()
    uv[~(z >= 0)] = np.nan
    return uv


def convex_edges(poly: np.ndarray, H: int, W: int):
    """kornia _get_convex_edges: per scan line y the left-most / right-most x of the active edges (W / -1 if none)."""
    poly = poly.astype(f32)
    a, b = poly[-1], poly[0]
    close = np.all(np.abs(a - b) <= f32(1e-8) + f32(1e-5) * np.abs(b))   # torch.allclose; NaN -> False
    if not close:
        poly = np.concatenate([poly, poly[:1]], axis=0)
    x0, y0, x1, y1 = poly[:-1, 0], poly[:-1, 1], poly[1:, 0], poly[1:, 1]
    ys = np.arange(H, dtype=f32)[:, None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        dx = ((x1 - x0) / ((y1 - y0) + f32(1e-12))).astype(f32)
        dx = np.minimum(np.maximum(dx, f32(-W)), f32(W))     # torch.clamp / np.minimum keep a NaN (inf - inf over an infinite vertex)
        xs = (((ys - y0[None]) * dx[None]).astype(f32) + x0[None]).astype(f32)
    act = ((y0[None] <= ys) & (ys <= y1[None])) | ((y0[None] >= ys) & (ys >= y1[None]))
    # torch's min / max over the edge axis propagate a NaN (np.min / np.max do too): such a scan line is never filled
    left = np.where(act, xs, f32(W)).min(axis=1).astype(f32) if xs.shape[1] else np.full(H, f32(W))
    right = np.where(act, xs, f32(-1)).max(axis=1).astype(f32) if xs.shape[1] else np.full(H, f32(-1))
    return left, right


def fill_mask(poly: np.ndarray, H: int, W: int) -> np.ndarray:
    """bool [H,W]: inside the convex polygon by kornia's rule  x_left <= x <= x_right."""
    left, right = convex_edges(poly, H, W)
    ws = np.arange(W, dtype=f32)[None]
    return (ws >= left[:, None]) & (ws <= right[:, None])


def render_fmin(mask: np.ndarray, K, pose, pts, value: float) -> np.ndarray:
    """One node of add_supervision_node: mask [C,H,W] (NaN = unlabeled) -> fmin(mask, inside ? value : NaN)."""
    C, H, W = mask.shape
    inside = fill_mask(project_points(K, pose, pts), H, W)
    out = mask.astype(f32).copy()
    new = np.fmin(out, f32(value))
    out[:, inside] = new[:, inside]
    return out
