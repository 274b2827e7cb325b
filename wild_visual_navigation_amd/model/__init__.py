from .simple_mlp import SimpleMLP
from .network_register import get_model

__all__ = ["SimpleMLP", "get_model"]
