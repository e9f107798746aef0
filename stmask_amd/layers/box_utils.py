"""Box arithmetic of the hot path on the MI355X -- mirrors the reference's layers/box_utils.py for the functions
SURVEY.md §2 row 10 marks in scope: decode (:238-283), jaccard (:37-88), center_size (:25-35), point_form (:12-22),
sanitize_coordinates(_hw) (:298-337), crop (:341-364), mask_iou (:435-447).

decode / jaccard / mask_iou run as hand-written HIP kernels (bit-exact vs the oracle); the tiny element-wise helpers
stay as torch ops in the reference's operand order (IEEE add / sub / mul / div are identical on CPU and GPU).
"""
import torch

from .. import ops


def point_form(boxes):
    return torch.cat((boxes[:, :2] - boxes[:, 2:] / 2, boxes[:, :2] + boxes[:, 2:] / 2), 1)


def center_size(boxes):
    return torch.cat(((boxes[:, 2:] + boxes[:, :2]) / 2, boxes[:, 2:] - boxes[:, :2]), 1)


def decode(loc, priors, use_yolo_regressors=False):
    if use_yolo_regressors:
        raise NotImplementedError("use_yolo_regressors is False in every STMask config (config.py)")
    if loc.shape[0] == 0:
        return loc.new_zeros(0, 4)
    return ops.decode(loc, priors)


def jaccard(box_a, box_b, iscrowd=False):
    if iscrowd:
        raise NotImplementedError("iscrowd is a training-only path")
    if box_a.dim() == 3:  # batched form used by per-class Fast NMS
        return torch.stack([ops.jaccard(a, b) for a, b in zip(box_a, box_b)])
    if box_a.shape[0] == 0 or box_b.shape[0] == 0:
        return box_a.new_zeros(box_a.shape[0], box_b.shape[0])
    return ops.jaccard(box_a, box_b)


def sanitize_coordinates(_x1, _x2, img_size, padding=0, cast=True):
    _x1 = _x1 * img_size
    _x2 = _x2 * img_size
    if cast:
        _x1, _x2 = _x1.long(), _x2.long()
    x1, x2 = torch.min(_x1, _x2), torch.max(_x1, _x2)
    return torch.clamp(x1 - padding, min=0), torch.clamp(x2 + padding, max=img_size)


def sanitize_coordinates_hw(box, h, w):
    squeeze = box.dim() == 2
    if squeeze:
        box = box[None]
    x1, x2 = sanitize_coordinates(box[:, :, 0], box[:, :, 2], w, cast=False)
    y1, y2 = sanitize_coordinates(box[:, :, 1], box[:, :, 3], h, cast=False)
    out = torch.stack([x1, y1, x2, y2], dim=-1)
    return out[0] if squeeze else out


def crop(masks, boxes, padding=1):
    """masks [h,w,n], boxes [n,4] relative -> (crop_mask, masks * crop_mask).  The fused kernel
    (ops.lincomb_sigmoid_crop) is what the hot path uses; this torch form exists for API parity."""
    h, w, n = masks.shape
    x1, x2 = sanitize_coordinates(boxes[:, 0], boxes[:, 2], w, padding, cast=False)
    y1, y2 = sanitize_coordinates(boxes[:, 1], boxes[:, 3], h, padding, cast=False)
    cols = torch.arange(w, device=masks.device, dtype=x1.dtype).view(1, -1, 1)
    rows = torch.arange(h, device=masks.device, dtype=x1.dtype).view(-1, 1, 1)
    crop_mask = ((cols >= x1.view(1, 1, -1)) & (cols < x2.view(1, 1, -1)) & (rows >= y1.view(1, 1, -1)) &
                 (rows < y2.view(1, 1, -1))).float()
    return crop_mask, masks * crop_mask


def mask_iou(mask1, mask2, thr=0.5):
    """[n1,h,w] x [n2,h,w] -> [n1,n2].  Inputs may be soft masks or already-binarised 0/1 floats (the reference passes
    m.gt(0.5).float(), track_TF.py:85,107): both binarise identically under `> 0.5`."""
    return ops.mask_iou(mask1, mask2, thr)
