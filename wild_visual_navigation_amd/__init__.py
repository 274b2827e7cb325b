"""wild_visual_navigation_amd -- MI355X (gfx950) native implementation of Wild Visual Navigation's
feature_extractor -> traversability_estimator hot path.  Mirrors the reference's Python API
(FeatureExtractor, DinoInterface, StegoInterface, SegmentExtractor, SimpleMLP/get_model, Data/Batch,
ConfidenceGenerator, TraversabilityLoss, TraversabilityEstimator); the arithmetic runs in the
hand-written HIP kernels of libwvn_hip.so (include/wvn_hip.h).  See INTEGRATION.md for the drop-in shim."""
import os

WVN_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
__version__ = "0.1.0"
