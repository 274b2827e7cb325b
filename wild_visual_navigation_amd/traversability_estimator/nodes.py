"""MissionNode -- the hot-path slice of wild_visual_navigation/traversability_estimator/nodes.py:116-440:
per-frame features / segments / supervision mask, label pooling (``update_supervision_signal``, HIP
kernel) and ``as_pyg_data``.  Pose bookkeeping, projection and debug imagery are out of scope."""
from typing import Optional

import torch

from .. import ops
from ..utils.data import Data


class MissionNode:
    _name = "mission_node"

    def __init__(self, timestamp: float = 0.0, camera_name: str = "cam", use_for_training: bool = True):
        self._timestamp = timestamp
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

    # -- properties the learning node sets after construction (wvn_learning_node.py:653-656) --
    features = property(lambda s: s._features, lambda s, v: setattr(s, "_features", v))
    feature_edges = property(lambda s: s._feature_edges, lambda s, v: setattr(s, "_feature_edges", v))
    feature_segments = property(lambda s: s._feature_segments, lambda s, v: setattr(s, "_feature_segments", v))
    feature_positions = property(lambda s: s._feature_positions, lambda s, v: setattr(s, "_feature_positions", v))
    prediction = property(lambda s: s._prediction, lambda s, v: setattr(s, "_prediction", v))
    supervision_mask = property(lambda s: s._supervision_mask, lambda s, v: setattr(s, "_supervision_mask", v))
    supervision_signal = property(lambda s: s._supervision_signal)
    supervision_signal_valid = property(lambda s: s._supervision_signal_valid)
    confidence = property(lambda s: s._confidence, lambda s, v: setattr(s, "_confidence", v))
    timestamp = property(lambda s: s._timestamp)
    use_for_training = property(lambda s: s._use_for_training)

    def update_supervision_signal(self):
        """nodes.py:400-440: nanmean over the mask channels, then per-segment mean of the labelled
        pixels; 0 where a segment has no label; valid = signal > 0."""
        if self._supervision_mask is None or self._features is None:
            return
        n_seg = int(self._features.shape[0])  # == feature_segments.max() + 1 for compacted ids
        self._supervision_signal, self._supervision_signal_valid = ops.label_pool(
            self._supervision_mask, self._feature_segments, n_seg)

    def as_pyg_data(self, previous_node=None, anomaly_detection: bool = False, aux: bool = False):
        if aux:
            return Data(x=self.features, edge_index=self._feature_edges)
        return Data(x=self.features, edge_index=self._feature_edges, y=self._supervision_signal,
                    y_valid=self._supervision_signal_valid)

    def is_valid(self):
        ok = all(isinstance(t, torch.Tensor) for t in (self._features, self._supervision_signal,
                                                       self._supervision_signal_valid))
        return bool(ok and self._supervision_signal_valid.any())
