from .data import Data, Batch
from .confidence_generator import ConfidenceGenerator
from .loss import TraversabilityLoss

__all__ = ["Data", "Batch", "ConfidenceGenerator", "TraversabilityLoss"]
