from .nodes import BaseNode, MissionNode, SupervisionNode
from .graphs import BaseGraph, DistanceWindowGraph, MaxElementsGraph, MissionGraph, TemporalWindowGraph
from .trainer import MlpTrainer
from .traversability_estimator import TraversabilityEstimator

__all__ = ["BaseNode", "MissionNode", "SupervisionNode", "BaseGraph", "DistanceWindowGraph", "MaxElementsGraph",
           "TemporalWindowGraph", "MissionGraph", "MlpTrainer", "TraversabilityEstimator"]
