"""Frame pre-processing, SURVEY.md section 8 row f3 -- the host chain of eval.py:703-717 (`evaluate_single`) as one device
kernel: resize to 640x360 (cv2 INTER_LINEAR semantics on uint8), normalise with config.py:27-28 MEANS / STD, zero-pad to
a multiple of 32, HWC -> CHW fp32.  Returns the tensor and the `img_meta` dict the reference builds next to it."""
import torch

from . import ops

MEANS = (123.675, 116.28, 103.53)   # datasets/config.py:27
STD = (58.395, 57.12, 57.375)       # datasets/config.py:28
_MODES = {"normalize": 1, "subtract_means": 2, "to_float": 3, None: 0}


def preprocess_eval_frames(frames_u8, idx=None, size=(640, 360), transform="normalize"):
    """frames_u8: uint8 [n, H, W, 3] on the GPU (the n clips' current frames, channel order as mmcv.imread gives it).
    -> (fp32 [n, 3, 384, 640], img_meta) with img_meta as evaluate_single builds it (eval.py:716-722)."""
    n, H0, W0, _ = frames_u8.shape
    out = ops.preprocess_frames(frames_u8, size=size, divisor=32, mean=MEANS, std=STD, mode=_MODES[transform])
    w, h = size
    meta = {"ori_shape": (H0, W0, 3), "img_shape": (h, w, 3), "pad_shape": (out.shape[2], out.shape[3], 3)}
    if idx is not None:
        meta["frame_id"] = idx
    meta["is_first"] = idx is None or idx == 0
    return out, meta
