from stmask_amd.mmcv_ops import DeformConv2d, RoIAlign, roi_align  # noqa: F401
