"""Drop-in installer: make ``import wild_visual_navigation...`` resolve to this package for the hot path.

    import wild_visual_navigation_amd.dropin as dropin; dropin.install()        # first lines of quick_start.py / a ROS node,
                                                                                  # or a sitecustomize.py

After ``install()`` the reference's own import lines (quick_start.py:6-18, wvn_feature_extractor_node.py:7-12,
wvn_learning_node.py:7-24) bind the MI355X classes:

    from wild_visual_navigation import WVN_ROOT_DIR
    from wild_visual_navigation.feature_extractor import FeatureExtractor           (DinoInterface, StegoInterface, SegmentExtractor)
    from wild_visual_navigation.image_projector import ImageProjector
    from wild_visual_navigation.model import get_model                              (SimpleMLP)
    from wild_visual_navigation.utils import ConfidenceGenerator, Data, Batch, TraversabilityLoss, AnomalyLoss, WVNMode, make_plane ...
    from wild_visual_navigation.traversability_estimator import TraversabilityEstimator, MissionNode, SupervisionNode, graphs
    from wild_visual_navigation.cfg import ExperimentParams

Two situations:
  * the reference package IS importable (a robot with WVN installed): only the hot-path sub-packages listed in ``HOT`` are
    replaced; everything else the callers import -- ``visu``, ``supervision_generator``, ``cfg.Ros*Params``,
    ``utils.create_experiment_folder`` -- keeps coming from the reference (those are CPU-side helpers outside this build's
    scope);
  * it is NOT (this container): a synthetic ``wild_visual_navigation`` package is registered whose sub-modules are the ones of
    this package, so the import lines above work and anything out of scope raises ImportError as usual.
"""
import importlib
import os
import sys
import types

HOT = ("feature_extractor", "model", "traversability_estimator", "image_projector", "cfg")
# names of wild_visual_navigation.utils that belong to the hot path (the rest of utils stays the reference's)
UTILS_HOT = ("Data", "Batch", "ConfidenceGenerator", "TraversabilityLoss", "AnomalyLoss", "WVNMode", "make_plane",
             "make_dense_plane", "make_polygon_from_points")


def _alias_submodules(pkg_mod, alias: str) -> None:
    """Register every sub-module of ``pkg_mod`` under the reference's dotted name too (``wild_visual_navigation.
    traversability_estimator.nodes`` -> the module object of ``wild_visual_navigation_amd.traversability_estimator.nodes``).
    Without this, ``from wild_visual_navigation.traversability_estimator.nodes import TwistNode`` (supervision_generator.py:8)
    would find nodes.py through the package's search path and execute it a SECOND time under the alias name -- different class
    objects, and its relative imports would resolve against the wrong parent."""
    import pkgutil

    for info in pkgutil.walk_packages(pkg_mod.__path__, prefix=pkg_mod.__name__ + "."):
        sub = importlib.import_module(info.name)
        sys.modules[alias + info.name[len(pkg_mod.__name__):]] = sub


def install(force_synthetic: bool = False) -> str:
    """Returns "overlay" (reference present, hot-path modules replaced) or "synthetic" (stand-alone alias package)."""
    import wild_visual_navigation_amd as amd

    mods = {name: importlib.import_module(f"wild_visual_navigation_amd.{name}") for name in HOT}
    amd_utils = importlib.import_module("wild_visual_navigation_amd.utils")
    ref = None
    if not force_synthetic:
        try:
            existing = sys.modules.get("wild_visual_navigation")
            ref = existing if existing is not None and not getattr(existing, "__wvn_amd_synthetic__", False) \
                else importlib.import_module("wild_visual_navigation")
            if getattr(ref, "__wvn_amd_synthetic__", False):
                ref = None
        except Exception:
            ref = None
    if ref is None:
        pkg = types.ModuleType("wild_visual_navigation")
        pkg.__path__ = []   # a package without a search path: only what is registered below can be imported from it
        pkg.__wvn_amd_synthetic__ = True
        pkg.WVN_ROOT_DIR = os.environ.get("WVN_ROOT_DIR", amd.WVN_ROOT_DIR)
        pkg.__doc__ = "alias of wild_visual_navigation_amd (hot path only)"
        sys.modules["wild_visual_navigation"] = pkg
        for name, m in mods.items():
            sys.modules[f"wild_visual_navigation.{name}"] = m
            setattr(pkg, name, m)
            _alias_submodules(m, f"wild_visual_navigation.{name}")
        sys.modules["wild_visual_navigation.utils"] = amd_utils
        pkg.utils = amd_utils
        _alias_submodules(amd_utils, "wild_visual_navigation.utils")
        return "synthetic"
    for name, m in mods.items():
        if name == "cfg":
            continue   # the reference's own config tree (OmegaConf dataclasses) works as is: the estimator reads it by key
        sys.modules[f"wild_visual_navigation.{name}"] = m
        setattr(ref, name, m)
        _alias_submodules(m, f"wild_visual_navigation.{name}")
    ref_utils = importlib.import_module("wild_visual_navigation.utils")
    for n in UTILS_HOT:
        setattr(ref_utils, n, getattr(amd_utils, n))
    return "overlay"
