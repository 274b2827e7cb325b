"""get_model -- wild_visual_navigation/model/network_register.py:44-55 (class-name registry; only the
default SimpleMLP is on the MI355X path)."""
from .simple_mlp import SimpleMLP

_REGISTER = {"SimpleMLP": (SimpleMLP, "simple_mlp_cfg")}


def _get(cfg, key):
    return cfg[key] if not hasattr(cfg, key) or isinstance(cfg, dict) else getattr(cfg, key)


def get_model(model_cfg):
    name = _get(model_cfg, "name")
    if name not in _REGISTER:
        raise KeyError(f"model '{name}' is not part of the MI355X hot path (available: {list(_REGISTER)})")
    cls, key = _REGISTER[name]
    sub = _get(model_cfg, key)
    kw = dict(sub) if isinstance(sub, dict) else {k: getattr(sub, k) for k in ("input_size", "hidden_sizes", "reconstruction")}
    kw["hidden_sizes"] = list(kw["hidden_sizes"])
    return cls(**kw)
