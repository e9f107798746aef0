"""`mmcv.ops` entry points of the reference's hot path on the MI355X kernels (Featurealign.py:3,27-31,72;
track_to_segment_head.py:6,86).  Other mmcv.ops names are not on the STMask hot path and are not provided."""
from stmask_amd.mmcv_ops import DeformConv2d, RoIAlign, roi_align  # noqa: F401


def __getattr__(name):
    raise AttributeError(f"mmcv.ops.{name} is not provided by the stmask_amd shim (hot path: DeformConv2d, roi_align, RoIAlign)")
