"""Run the host-side mirror (stmask_amd.model / layers) on the CPU by swapping the HIP-backed ``stmask_amd.ops``
entry points for the oracle.  TEST INFRASTRUCTURE ONLY: used by tests/ (host-logic parity against the reference
goldens without a GPU) and by bench.py's ``cpu_baseline`` leg.  The product never imports this module; outside the
``oracle_ops()`` context every stmask_amd op still refuses CPU tensors.
"""
import contextlib

import torch

from . import oracle as orc


def _deform_conv(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, deform_groups=1, relu=False,
                 fused_om=None, mask_is_logit=False):
    if fused_om is not None:
        K3 = fused_om.shape[1]
        offset, mask = fused_om[:, : 2 * K3 // 3], torch.sigmoid(fused_om[:, 2 * K3 // 3:])
    elif mask is not None and mask_is_logit:
        mask = torch.sigmoid(mask)
    y = orc.deform_conv(x, offset, mask, weight, bias, stride, padding, dilation, deform_groups)
    return torch.relu(y) if relu else y


def _generate_candidates(loc, priors, conf, thresh=0.05):
    B, N, _ = conf.shape
    keep_idx = torch.zeros(B, N, dtype=torch.int64)
    cand_box = torch.zeros(B, N, 4)
    count = torch.zeros(B, dtype=torch.int32)
    for b in range(B):
        k = orc.candidate_filter(conf[b], thresh)
        keep_idx[b, : len(k)] = k
        cand_box[b, : len(k)] = orc.decode(loc[b], priors)[k]
        count[b] = len(k)
    return keep_idx, cand_box, count


def _pad(n, *ts):
    out = []
    for t in ts:
        p = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype)
        p[: len(t)] = t
        out.append(p)
    return out


def _cc_fast_nms(conf, boxes, centerness, iou_thr=0.5, top_k=200, k_dev=None):
    idx, cls, sc = orc.cc_fast_nms(conf, boxes, centerness, iou_thr, top_k)
    n = len(idx)
    idx_p, cls_p, sc_p, bx_p = _pad(top_k, idx, cls, sc, boxes[idx])
    return idx_p, cls_p, sc_p, bx_p, torch.tensor(n, dtype=torch.int32)


def _fast_nms(conf, boxes, centerness, iou_thr=0.5, top_k=200, conf_thresh=0.05, max_det=100, k_dev=None):
    idx, cls, sc = orc.fast_nms(conf, boxes, centerness, iou_thr, top_k, conf_thresh, max_det)
    n = len(idx)
    idx_p, cls_p, sc_p, bx_p = _pad(max_det, idx, cls, sc, boxes[idx])
    return idx_p, cls_p, sc_p, bx_p, torch.tensor(n, dtype=torch.int32)


def _detect_cc(loc, priors, conf, centerness, conf_thresh=0.05, iou_thr=0.5, top_k=200):
    B, N, _ = conf.shape
    outs = []
    for b in range(B):
        keep = orc.candidate_filter(conf[b], conf_thresh)
        boxes = orc.decode(loc[b], priors)
        cen = centerness[b].reshape(-1)[keep] if centerness is not None else None
        idx, cls, sc = orc.cc_fast_nms(conf[b][keep], boxes[keep], cen, iou_thr, top_k)
        pri_idx = keep[idx]
        outs.append(_pad(top_k, pri_idx, cls, sc, boxes[pri_idx]) + [torch.tensor(len(idx), dtype=torch.int32)])
    return tuple(torch.stack([o[i] for o in outs]) for i in range(5))


def _lincomb(proto, coeff, boxes=None, apply_tanh=True, n_dev=None, row_proto=None):
    if row_proto is not None:
        out = torch.zeros(coeff.shape[0], proto.shape[1], proto.shape[2])
        for pi in row_proto.unique().tolist():
            sel = torch.nonzero(row_proto == pi).view(-1)
            out[sel] = orc.generate_mask(proto[pi], coeff[sel], boxes[sel] if boxes is not None else None, apply_tanh)
    else:
        out = orc.generate_mask(proto, coeff, boxes, apply_tanh)
    if n_dev is not None:
        out[int(n_dev):] = 0
    return out


def _bias_act_(y, bias, residual=None, relu=True):
    y.add_(bias.view(1, -1, 1, 1))
    if residual is not None:
        y.add_(residual)
    if relu:
        y.clamp_(min=0)
    return y


def _mask_iou(m1, m2, thr=0.5, group1=None, group2=None):
    """ops.mask_iou's contract: with groups, only pairs of the same group (clip) are computed, the rest are 0."""
    iou = orc.mask_iou(m1, m2, thr)
    if group1 is not None:
        iou = iou * (group1.view(-1, 1) == group2.view(1, -1)).to(iou.dtype)
    return iou


_PATCH = {
    "bias_act_": _bias_act_,
    "deform_conv": _deform_conv,
    "deform_im2col": lambda x, offset, mask, kernel_size, stride=1, padding=0, dilation=1, deform_groups=1, variant=0,
    fused_om=None, mask_is_logit=False, out=None: orc.deform_im2col(x, offset, mask, kernel_size, stride, padding, dilation,
                                                                    deform_groups),
    "fcb_ali_offsets": orc.fcb_ali_offsets,
    "corr_patch": lambda f1, f2, patch_size=11, dilation_patch=1, scale=1.0, leaky_slope=1.0: orc.corr_patch(
        f1, f2, patch_size, dilation_patch, scale, leaky_slope),
    "roi_align": lambda feat, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True: orc.roi_align(
        feat, rois, output_size, spatial_scale, sampling_ratio, "avg", aligned),
    "decode": orc.decode,
    "generate_candidates": _generate_candidates,
    "cc_fast_nms": _cc_fast_nms,
    "fast_nms": _fast_nms,
    "detect_cc": _detect_cc,
    "jaccard": orc.jaccard,
    "lincomb_sigmoid_crop": _lincomb,
    "mask_iou": _mask_iou,
}


@contextlib.contextmanager
def oracle_ops():
    """Inside the context, stmask_amd.ops.* are served by the CPU oracle (for CPU parity tests / the CPU baseline)."""
    from stmask_amd import ops
    saved = {k: getattr(ops, k) for k in _PATCH}
    try:
        for k, v in _PATCH.items():
            setattr(ops, k, v)
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
