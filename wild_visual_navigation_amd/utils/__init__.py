from .data import Data, Batch
from .confidence_generator import ConfidenceGenerator
from .loss import TraversabilityLoss, AnomalyLoss
from .meshes import make_plane, make_dense_plane, make_polygon_from_points
from .operation_modes import WVNMode

__all__ = ["Data", "Batch", "ConfidenceGenerator", "TraversabilityLoss", "AnomalyLoss", "make_plane", "make_dense_plane",
           "make_polygon_from_points", "WVNMode"]
