"""ImageProjector -- wild_visual_navigation/image_projector/image_projector.py:16-200 with the same constructor, ``camera``
attribute, ``project``, ``project_and_render`` and ``resize_image``.  The projection + convex-polygon fill of
``project_and_render`` (kornia PinholeCamera.project + draw_convex_polygon in the reference) is the HIP kernel of
csrc/supervision.hip; ``TraversabilityEstimator.add_supervision_node`` calls the same kernel in its fused, in-place form."""
import torch

from .. import ops
from ..feature_extractor.transforms import ingest_tables, resize_nearest_center_crop


class PinholeCamera:
    """The slice of kornia.geometry.camera.PinholeCamera the path reads: batched intrinsics / extrinsics [B,4,4], height,
    width, batch_size, camera_matrix."""

    def __init__(self, intrinsics: torch.Tensor, extrinsics: torch.Tensor, height: torch.Tensor, width: torch.Tensor):
        self.intrinsics, self.extrinsics, self.height, self.width = intrinsics, extrinsics, height, width

    @property
    def batch_size(self) -> int:
        return self.intrinsics.shape[0]

    @property
    def camera_matrix(self) -> torch.Tensor:
        return self.intrinsics[..., :3, :3]


class ImageProjector:
    def __init__(self, K: torch.Tensor, h, w, new_h: int = None, new_w: int = None):
        device = K.device
        E = torch.eye(4).expand(K.shape).to(device)
        self.K = K
        self.height = h
        self.width = w
        h_val = float(h.item() if isinstance(h, torch.Tensor) else h)
        w_val = float(w.item() if isinstance(w, torch.Tensor) else w)
        new_h = int(h_val) if new_h is None else new_h
        sy = new_h / h_val
        sx = (new_w / w_val) if (new_w is not None) else sy
        sh = new_h
        sw = new_w if new_w is not None else sh
        self._crop = (new_h, None) if (new_w is None or new_w == new_h) else (new_h, new_w)
        sK = K.clone()
        if new_w is None or new_w == new_h:   # image_projector.py:64-68: the square crop scales BOTH axes by sy and uses (fy, cy) for x
            sK[:, 0, 0] = K[:, 1, 1] * sy
            sK[:, 0, 2] = K[:, 1, 2] * sy
            sK[:, 1, 1] = K[:, 1, 1] * sy
            sK[:, 1, 2] = K[:, 1, 2] * sy
        else:
            sK[:, 0, 0] = K[:, 0, 0] * sx
            sK[:, 0, 2] = K[:, 0, 2] * sx
            sK[:, 1, 1] = K[:, 1, 1] * sy
            sK[:, 1, 2] = K[:, 1, 2] * sy
        self.camera = PinholeCamera(sK, E, torch.IntTensor([sh]).to(device), torch.IntTensor([sw]).to(device))
        self.masks = None

    @property
    def scaled_camera_matrix(self):
        return self.camera.intrinsics.clone()[:3, :3]

    def change_device(self, device):
        self.K = self.K.to(device)
        c = self.camera
        self.camera = PinholeCamera(c.intrinsics.to(device), c.extrinsics.to(device), c.height.to(device), c.width.to(device))

    def check_validity(self, points_3d: torch.Tensor, points_2d: torch.Tensor):
        valid_z = points_3d[..., 2] >= 0
        valid = (valid_z & (points_2d[..., 0] >= 0) & (points_2d[..., 0] <= self.camera.width)
                 & (points_2d[..., 1] >= 0) & (points_2d[..., 1] <= self.camera.height))
        return valid, valid_z

    def _render(self, pose_camera_in_world, points, masks, value):
        B = self.camera.batch_size
        Ks = [self.camera.intrinsics[i] for i in range(B)]
        poses = [pose_camera_in_world[i] for i in range(B)]
        return ops.project_render_fmin(Ks, poses, masks, points, value, want_projected=True)

    def project(self, pose_camera_in_world: torch.Tensor, points_W: torch.Tensor):
        """image_projector.py:126-150 -> (projected [B,N,2] raw pinhole coordinates -- finite also for points behind the
        camera, as in the reference --, valid [B,N], valid_z [B,N] = camera-frame z >= 0)."""
        B = self.camera.batch_size
        H, W = int(self.camera.height.item()), int(self.camera.width.item())
        scratch = [torch.full((1, H, W), float("nan"), device=points_W.device) for _ in range(B)]
        proj, depth = self._render(pose_camera_in_world, points_W, scratch, 1.0)
        valid_z = depth >= 0
        valid = (valid_z & (proj[..., 0] >= 0) & (proj[..., 0] <= self.camera.width) & (proj[..., 1] >= 0)
                 & (proj[..., 1] <= self.camera.height))
        return proj, valid, valid_z

    def project_and_render(self, pose_camera_in_world: torch.Tensor, points: torch.Tensor, colors: torch.Tensor,
                           image: torch.Tensor = None):
        """image_projector.py:152-197 -> (masks [B,3,H,W] with NaN outside the polygon, image_overlay, projected, valid).
        ``colors``: [3] or [B,3]; as in the reference's only caller a uniform colour per batch element is filled."""
        B = self.camera.batch_size
        H, W = int(self.camera.height.item()), int(self.camera.width.item())
        dev = points.device
        colors = colors.to(dev).float()
        if colors.dim() == 1:
            colors = colors[None].expand(B, 3)
        cover = [torch.full((1, H, W), float("nan"), dtype=torch.float32, device=dev) for _ in range(B)]
        proj, depth = self._render(pose_camera_in_world, points, cover, 1.0)   # ONE launch: inside = 1, outside stays NaN
        proj = torch.where((depth >= 0)[..., None], proj, torch.full((), float("nan"), device=dev))   # :180
        inside = ~torch.isnan(torch.stack(cover))                                # [B,1,H,W]
        self.masks = torch.where(inside, colors[:, :, None, None].expand(B, 3, H, W), torch.zeros((), device=dev))
        self.masks[self.masks == 0.0] = float("nan")
        overlay = image
        if image is not None:
            img = image if image.dim() == 4 else image[None]
            overlay = torch.where(inside, colors[:, :, None, None].expand_as(img), img)
        valid_z = depth >= 0
        valid = (valid_z & (proj[..., 0] >= 0) & (proj[..., 0] <= self.camera.width) & (proj[..., 1] >= 0)
                 & (proj[..., 1] <= self.camera.height))
        return self.masks, overlay, proj, valid

    def resize_image(self, image: torch.Tensor):
        """image_projector.py:199-200 (T.Resize(NEAREST) + T.CenterCrop, or a plain NEAREST resize to [new_h, new_w])."""
        nh, nw = self._crop
        H, W = image.shape[-2:]
        if image.is_cuda and image.element_size() in (1, 4):   # one HIP gather through the geometry's index tables
            return ops.resize_nearest_crop(image, ingest_tables(H, W, nh, image.device, out_w=nw))
        x = image if image.dim() == 4 else image[None]   # host tensors / other dtypes: the torch restatement (plumbing)
        if nw is None:
            out = resize_nearest_center_crop(x, nh)
        else:
            out = torch.nn.functional.interpolate(x.float(), size=(nh, nw), mode="nearest").to(image.dtype)
        return out if image.dim() == 4 else out[0]
