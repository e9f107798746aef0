"""Same public names as the reference's ``layers`` package (layers/__init__.py:1-2, functions/__init__.py:1-9,
modules/__init__.py:1-12) for the components on the hot path."""
from .box_utils import center_size, crop, decode, jaccard, mask_iou, point_form, sanitize_coordinates, \
    sanitize_coordinates_hw  # noqa: F401
from .functions import CandidateShift, Detect, Detect_TF, Track, Track_TF, compute_comp_scores, generate_candidate, \
    merge_candidates  # noqa: F401
from .mask_utils import generate_mask  # noqa: F401
from .modules import FPN, FeatureAlign, InterpolateModule, PredictionModule_FC, TemporalNet, bbox_feat_extractor, \
    correlate, make_net  # noqa: F401
