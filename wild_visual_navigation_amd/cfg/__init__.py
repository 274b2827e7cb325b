from .experiment_params import ExperimentParams

__all__ = ["ExperimentParams"]
