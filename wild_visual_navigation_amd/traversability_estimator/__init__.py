from .nodes import BaseNode, MissionNode, SupervisionNode, TwistNode, run_base_state
from .graphs import (BaseGraph, DistanceWindowGraph, MaxElementsGraph, MissionGraph, TemporalWindowGraph, run_base_graph,
                     run_temporal_window_graph)
from .trainer import MlpTrainer
from .traversability_estimator import TraversabilityEstimator

__all__ = ["BaseNode", "MissionNode", "SupervisionNode", "BaseGraph", "DistanceWindowGraph", "MaxElementsGraph",
           "TemporalWindowGraph", "MissionGraph", "MlpTrainer", "TraversabilityEstimator", "TwistNode", "run_base_state",
           "run_base_graph", "run_temporal_window_graph"]
