"""Mirror of wild_visual_navigation.feature_extractor (same class names / signatures), backed by
the HIP library.  See INTEGRATION.md for the import shim."""
from .dino_interface import DinoInterface
from .stego_interface import StegoInterface
from .segment_extractor import SegmentExtractor
from .feature_extractor import FeatureExtractor

__all__ = ["DinoInterface", "StegoInterface", "SegmentExtractor", "FeatureExtractor"]
