"""Graph nodes -- wild_visual_navigation/traversability_estimator/nodes.py:20-617: ``BaseNode`` (pose, ordering, SE(3)
distance), ``MissionNode`` (per-frame features / segments / supervision mask, label pooling on the HIP kernel, ``as_pyg_data``)
and ``SupervisionNode`` (footprint geometry feeding the supervision-mask rasteriser).  Constructors, properties and method
names follow the reference; debug imagery / dataset export (``save``, ``clear_debug_data`` internals) are reduced to what the
path needs."""
from typing import Optional

import torch

from .. import ops
from ..utils.data import Data
from ..utils.meshes import make_dense_plane, make_plane, make_polygon_from_points
from ..utils.se3 import se3_log_translation_norm, so3_from_rpy


class BaseNode:
    """nodes.py:20-113."""

    _name = "base_node"

    def __init__(self, timestamp: float = 0.0, pose_base_in_world: torch.Tensor = torch.eye(4)):
        assert isinstance(pose_base_in_world, torch.Tensor)
        self._timestamp = timestamp
        self._pose_base_in_world = pose_base_in_world

    def __str__(self):
        return f"{self._name}_{self._timestamp}"

    def __hash__(self):
        return hash(str(self))

    def __eq__(self, other):
        if other is None:
            return False
        return (self._name == other.name and self._timestamp == other.timestamp
                and torch.equal(self._pose_base_in_world, other.pose_base_in_world))

    def __lt__(self, other):
        return self._timestamp < other.timestamp

    def change_device(self, device):
        self._pose_base_in_world = self._pose_base_in_world.to(device)

    @classmethod
    def from_node(cls, instance):
        return cls(timestamp=instance.timestamp, pose_base_in_world=instance.pose_base_in_world)

    def is_valid(self):
        return True

    def pose_between(self, other):
        return other.pose_base_in_world.inverse() @ self.pose_base_in_world

    def distance_to(self, other):
        """nodes.py:73-91: norm of the translational part of log(T_self^-1 T_other) on SE(3)."""
        rel = self.pose_base_in_world.inverse() @ other.pose_base_in_world.to(self.pose_base_in_world.device)
        return se3_log_translation_norm(rel.detach().float().cpu())

    name = property(lambda s: s._name)
    timestamp = property(lambda s: s._timestamp, lambda s, v: setattr(s, "_timestamp", v))
    pose_base_in_world = property(lambda s: s._pose_base_in_world, lambda s, v: setattr(s, "_pose_base_in_world", v))


class MissionNode(BaseNode):
    """nodes.py:116-440: everything is stored on the image plane."""

    _name = "mission_node"

    def __init__(self, timestamp: float = 0.0, pose_base_in_world: torch.Tensor = torch.eye(4),
                 pose_cam_in_base: torch.Tensor = torch.eye(4), pose_cam_in_world: torch.Tensor = None,
                 image: torch.Tensor = None, image_projector=None, camera_name: str = "cam", use_for_training: bool = True):
        super().__init__(timestamp=timestamp, pose_base_in_world=pose_base_in_world)
        self._pose_cam_in_base = pose_cam_in_base
        self._pose_cam_in_world = (self._pose_base_in_world @ self._pose_cam_in_base.to(self._pose_base_in_world.device)
                                   if pose_cam_in_world is None else pose_cam_in_world)
        self._image = image
        self._image_projector = image_projector
        self._camera_name = camera_name
        self._use_for_training = use_for_training
        self._features = None
        self._feature_edges = None
        self._feature_segments = None
        self._feature_positions = None
        self._prediction = None
        self._supervision_mask = None
        self._supervision_signal = None
        self._supervision_signal_valid = None
        self._confidence = None
        self._seg_i32 = None   # int32 copy of feature_segments for the kernels (the reference keeps int64)
        self._num_segments = None

    def clear_debug_data(self):
        """nodes.py:152-162: drop what training does not need."""
        self._image = None
        self._supervision_mask = None

    def change_device(self, device):
        super().change_device(device)
        if self._image_projector is not None:
            self._image_projector.change_device(device)
        for n in ("_pose_cam_in_base", "_pose_cam_in_world", "_image", "_features", "_feature_edges", "_feature_segments",
                  "_feature_positions", "_prediction", "_supervision_mask", "_supervision_signal",
                  "_supervision_signal_valid", "_confidence", "_seg_i32"):
            t = getattr(self, n)
            if t is not None:
                setattr(self, n, t.to(device))

    # -- properties the learning node sets after construction (wvn_learning_node.py:653-656) --
    features = property(lambda s: s._features, lambda s, v: setattr(s, "_features", v))
    feature_edges = property(lambda s: s._feature_edges, lambda s, v: setattr(s, "_feature_edges", v))
    feature_positions = property(lambda s: s._feature_positions, lambda s, v: setattr(s, "_feature_positions", v))
    prediction = property(lambda s: s._prediction, lambda s, v: setattr(s, "_prediction", v))
    supervision_mask = property(lambda s: s._supervision_mask, lambda s, v: setattr(s, "_supervision_mask", v))
    supervision_signal = property(lambda s: s._supervision_signal)
    supervision_signal_valid = property(lambda s: s._supervision_signal_valid)
    confidence = property(lambda s: s._confidence, lambda s, v: setattr(s, "_confidence", v))
    use_for_training = property(lambda s: s._use_for_training)
    camera_name = property(lambda s: s._camera_name)
    image = property(lambda s: s._image, lambda s, v: setattr(s, "_image", v))
    image_projector = property(lambda s: s._image_projector, lambda s, v: setattr(s, "_image_projector", v))
    pose_cam_in_world = property(lambda s: s._pose_cam_in_world, lambda s, v: setattr(s, "_pose_cam_in_world", v))
    pose_cam_in_base = property(lambda s: s._pose_cam_in_base, lambda s, v: setattr(s, "_pose_cam_in_base", v))

    @property
    def feature_segments(self):
        return self._feature_segments

    @feature_segments.setter
    def feature_segments(self, v):
        self._feature_segments = v
        self._seg_i32 = None
        self._num_segments = None

    def segments_i32(self) -> torch.Tensor:
        if self._seg_i32 is None:
            self._seg_i32 = self._feature_segments.to(torch.int32).contiguous()
        return self._seg_i32

    def num_segments(self) -> int:
        """Length of the supervision signal: ``feature_segments.max() + 1`` as in the reference (nodes.py:413); one host sync
        per node (cached: the map of a node does not change), never the feature-row count -- a map whose ids exceed the
        feature rows would otherwise lose those pixels silently."""
        if self._num_segments is None:
            self._num_segments = int(self._feature_segments.max().item()) + 1
        return self._num_segments

    def update_supervision_signal(self):
        """nodes.py:400-440: nanmean over the mask channels, then per-segment mean of the labelled pixels; 0 where a
        segment has no label; valid = signal > 0."""
        if self._supervision_mask is None or self._features is None:
            return
        self._supervision_signal, self._supervision_signal_valid = ops.label_pool(
            self._supervision_mask, self.segments_i32(), self.num_segments())

    def as_pyg_data(self, previous_node: Optional[BaseNode] = None, anomaly_detection: bool = False, aux: bool = False):
        """nodes.py:199-241."""
        if aux:
            return Data(x=self.features, edge_index=self._feature_edges)
        if anomaly_detection:
            v = self._supervision_signal_valid
            d = dict(x=self.features[v], edge_index=self._feature_edges, y=self._supervision_signal[v], y_valid=v[v])
        else:
            d = dict(x=self.features, edge_index=self._feature_edges, y=self._supervision_signal,
                     y_valid=self._supervision_signal_valid)
        if previous_node is not None:
            d.update(x_previous=previous_node.features, edge_index_previous=previous_node._feature_edges)
        return Data(**d)

    def is_valid(self):
        ok = all(isinstance(t, torch.Tensor) for t in (self._features, self._supervision_signal,
                                                       self._supervision_signal_valid))
        return bool(ok and self._supervision_signal_valid.any())


class SupervisionNode(BaseNode):
    """nodes.py:443-617: a robot pose with its footprint and the traversability measured there."""

    _name = "supervision_node"

    def __init__(self, timestamp: float = 0.0, pose_base_in_world: torch.Tensor = torch.eye(4),
                 pose_footprint_in_base: torch.Tensor = torch.eye(4), pose_footprint_in_world: torch.Tensor = None,
                 twist_in_base: torch.Tensor = None, desired_twist_in_base: torch.Tensor = None, length: float = 0.1,
                 width: float = 0.1, height: float = 0.1, supervision: torch.Tensor = None,
                 traversability: torch.Tensor = torch.FloatTensor([0.0]),
                 traversability_var: torch.Tensor = torch.FloatTensor([1.0]), is_untraversable: bool = False):
        assert isinstance(pose_base_in_world, torch.Tensor) and isinstance(pose_footprint_in_base, torch.Tensor)
        super().__init__(timestamp=timestamp, pose_base_in_world=pose_base_in_world)
        self._pose_footprint_in_base = pose_footprint_in_base
        self._pose_footprint_in_world = (self._pose_base_in_world @ self._pose_footprint_in_base.to(self._pose_base_in_world.device)
                                         if pose_footprint_in_world is None else pose_footprint_in_world)
        self._twist_in_base = twist_in_base
        self._desired_twist_in_base = desired_twist_in_base
        self._length, self._width, self._height = length, width, height
        self._supervision_state = supervision
        self._traversability = traversability
        self._traversability_var = traversability_var
        self._is_untraversable = is_untraversable

    def change_device(self, device):
        super().change_device(device)
        for n in ("_pose_footprint_in_base", "_pose_footprint_in_world", "_twist_in_base", "_desired_twist_in_base",
                  "_supervision_state"):
            t = getattr(self, n)
            if t is not None:
                setattr(self, n, t.to(device))

    def get_footprint_points(self):
        return make_plane(x=self._length, y=self._width, pose=self._pose_footprint_in_world, grid_size=25)

    def get_side_points(self):
        return make_plane(x=0.0, y=self._width, pose=self._pose_footprint_in_world, grid_size=2)

    def get_untraversable_plane(self, grid_size=5):
        """nodes.py:512-548: a wall in front of the robot, perpendicular to the motion direction."""
        device = self._pose_footprint_in_world.device
        d = (self._twist_in_base / self._twist_in_base.norm()).cpu()
        z_angle = torch.atan2(d[1], d[0]).item()
        rho = torch.tensor([0.5 * self._length * float(d[0]), 0.5 * self._length * float(d[1]), -self._height / 2])
        T = torch.eye(4)
        T[:3, :3] = so3_from_rpy(0.0, 0.0, z_angle)
        T[:3, 3] = rho
        pose_plane_in_world = self._pose_base_in_world @ T.to(device)
        return make_dense_plane(y=0.5 * self._width, z=self._height, pose=pose_plane_in_world, grid_size=grid_size)

    def make_footprint_with_node(self, other: "SupervisionNode", grid_size: int = 10):
        """nodes.py:550-571: quadrilateral spanned by this node's and the previous node's side points."""
        if self.is_untraversable:
            return self.get_untraversable_plane(grid_size=grid_size)
        other_side_points = other.get_side_points()
        this_side_points = self.get_side_points()
        this_side_points[[0, 1]] = this_side_points[[1, 0]]   # counter-clockwise
        points = torch.concat((this_side_points, other_side_points.to(this_side_points.device)), dim=0)
        return make_polygon_from_points(points, grid_size=grid_size)

    def update_traversability(self, traversability: torch.Tensor, traversability_var: torch.Tensor):
        if (traversability < self._traversability).any():   # pessimistic rule
            self._traversability = traversability
            self._traversability_var = traversability_var

    traversability = property(lambda s: s._traversability, lambda s, v: setattr(s, "_traversability", v))
    traversability_var = property(lambda s: s._traversability_var, lambda s, v: setattr(s, "_traversability_var", v))
    twist_in_base = property(lambda s: s._twist_in_base)
    desired_twist_in_base = property(lambda s: s._desired_twist_in_base)
    is_untraversable = property(lambda s: s._is_untraversable)
    pose_footprint_in_world = property(lambda s: s._pose_footprint_in_world)
    supervision_state = property(lambda s: s._supervision_state)

    def is_valid(self):
        return isinstance(self._supervision_state, torch.Tensor)


class TwistNode(BaseNode):
    """nodes.py:620-664: a pose with the desired and the measured twist (used by the reference's supervision generator, which
    keeps importing it from this module when only the hot-path packages are overlaid -- dropin.py)."""

    _name = "twist_node"

    def __init__(self, timestamp: float = 0.0, pose_base_in_world: torch.Tensor = torch.eye(4),
                 desired_twist: torch.Tensor = torch.zeros(6), current_twist: torch.Tensor = torch.zeros(6)):
        assert isinstance(pose_base_in_world, torch.Tensor) and isinstance(desired_twist, torch.Tensor) \
            and isinstance(current_twist, torch.Tensor)
        super().__init__(timestamp=timestamp, pose_base_in_world=pose_base_in_world)
        self._desired_twist = desired_twist
        self._current_twist = current_twist

    def change_device(self, device):
        super().change_device(device)
        self._desired_twist = self._desired_twist.to(device)
        self._current_twist = self._current_twist.to(device)

    desired_twist = property(lambda s: s._desired_twist, lambda s, v: setattr(s, "_desired_twist", v))
    current_twist = property(lambda s: s._current_twist, lambda s, v: setattr(s, "_current_twist", v))


def _translated(x: float) -> torch.Tensor:
    T = torch.eye(4)
    T[0, 3] = x
    return T


def run_base_state():
    """The reference's self-check of the node algebra (nodes.py:667-686, called by its tests/test_traversability_estimator.py):
    two states one metre and one second apart."""
    a, b = BaseNode(1, pose_base_in_world=_translated(1.0)), BaseNode(2, pose_base_in_world=_translated(2.0))
    assert abs(float(b.distance_to(a)) - 1.0) < 1e-6
    assert a != b
    assert b.timestamp - a.timestamp == 1.0
    assert BaseNode.from_node(a) == a
