from .nodes import MissionNode
from .graphs import MissionGraph
from .trainer import MlpTrainer
from .traversability_estimator import TraversabilityEstimator

__all__ = ["MissionNode", "MissionGraph", "MlpTrainer", "TraversabilityEstimator"]
