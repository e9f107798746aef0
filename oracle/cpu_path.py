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


def _detect_cc(loc, priors, conf, centerness, conf_thresh=0.05, iou_thr=0.5, top_k=200, logits=False):
    if logits:                      # the batched pipeline hands over raw class logits (STMask.py:314's softmax is folded into the kernel)
        conf = torch.softmax(conf, -1)
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


# ---- tracker bookkeeping (stmask_amd/csrc/tracker.hip): the torch chains those kernels replace -------------------------
def _gather_detections(idx, cls, score, box, cnt, mask_coeff, track, centerness, D):
    B, top_k = idx.shape
    N = mask_coeff.shape[1]
    valid = torch.arange(top_k)[None, :] < cnt[:, None]
    flat = (idx + torch.arange(B, dtype=torch.int64)[:, None] * N)[valid]
    return {"box": box[valid], "class": cls[valid], "score": score[valid],
            "mask_coeff": mask_coeff.reshape(B * N, -1).index_select(0, flat), "track": track.reshape(B * N, -1).index_select(0, flat),
            "centerness": centerness.reshape(B * N).index_select(0, flat),
            "clip": torch.repeat_interleave(torch.arange(B, dtype=torch.int32), cnt.long())}


def _shift_rois(box, clip, feat_h, feat_w):
    from stmask_amd.layers.box_utils import sanitize_coordinates_hw
    return torch.cat([clip.float().unsqueeze(1), sanitize_coordinates_hw(box, feat_h, feat_w)], dim=1)


def _shift_apply_(loc_shift, coeff_shift, box, mask_coeff, score, decay=0.95):
    from stmask_amd.layers.box_utils import center_size
    box.copy_(orc.decode(loc_shift.contiguous(), center_size(box)))
    mask_coeff.add_(coeff_shift)
    score.mul_(decay)


def _match_scores(cos, miou, det_box, prev_box, det_score, det_cls, prev_cls, det_clip, prev_offsets, match_coeff, dummy_iou=0.3):
    D = det_box.shape[0]
    c = match_coeff
    prev_clip = torch.repeat_interleave(torch.arange(prev_offsets.numel() - 1, dtype=torch.int32), (prev_offsets[1:] - prev_offsets[:-1]).long())
    cosf = (torch.cat([cos.new_zeros(D, 1), cos], dim=1) + 1) / 2
    dummy = torch.full((D, 1), dummy_iou)
    comp = cosf + c[0] * det_score.view(-1, 1) + c[1] * torch.cat([dummy, miou], 1) + c[2] * torch.cat([dummy, orc.jaccard(det_box, prev_box)], 1) \
        + c[3] * torch.cat([torch.ones_like(dummy), (prev_cls[None, :] == det_cls[:, None]).float()], 1)
    same = torch.cat([torch.ones(D, 1, dtype=torch.bool), det_clip[:, None] == prev_clip[None, :]], dim=1)
    return torch.where(same, comp, torch.full_like(comp, float("-inf"))).argmax(dim=1).to(torch.int32)


def _match_scores_embed(det_track, prev_track, miou, *rest, **kw):
    return _match_scores(det_track @ prev_track.t(), miou, *rest, **kw)     # track_TF.py:99-101


def _gather_rows2(a_rows, b_rows, plan, n_a):
    return [torch.cat([a, b], dim=0).index_select(0, plan.long()) for a, b in zip(a_rows, b_rows)]


def _pack_tracked(mask, score, tracked, offsets, box, cls, mask_coeff, B, top_k, cols, max_age=10, score_thr=0.05):
    out = torch.zeros(B, top_k, cols)
    if box.shape[0] == 0:
        return out
    keep = (tracked <= max_age) & (mask.gt(0.5).sum([1, 2]) > 1) & (score > score_thr)
    off = offsets.tolist()
    for b in range(B):
        rows = torch.nonzero(keep[off[b]:off[b + 1]]).view(-1)[:top_k]
        g = rows + off[b]
        n = rows.numel()
        out[b, :n, 0:4], out[b, :n, 4], out[b, :n, 5], out[b, :n, 6], out[b, :n, 7] = box[g], score[g], cls[g].float(), rows.float(), 1.0
        out[b, :n, 8:8 + mask_coeff.shape[1]] = mask_coeff[g]
    return out


def _lincomb_bits(proto, coeff, boxes, row_proto, thr=0.5, apply_tanh=True):
    """CPU stand-in: the soft masks plus a 'bit table' that simply is the binarised mask (only _mask_iou_bits reads it)."""
    out = _lincomb(proto, coeff, boxes, apply_tanh, None, row_proto)
    return out, out.gt(thr).reshape(out.shape[0], -1)


def _mask_iou_bits(bits1, bits2, hw, group1=None, group2=None):
    return _mask_iou(bits1.float().reshape(bits1.shape[0], 1, -1), bits2.float().reshape(bits2.shape[0], 1, -1), 0.5, group1, group2)


def _pack_tracked_bits(bits, score, tracked, offsets, box, cls, mask_coeff, B, top_k, cols, max_age=10, score_thr=0.05):
    # (the CPU stand-in of the bit table is the binarised mask itself: see _lincomb_bits)
    return _pack_tracked(bits.reshape(bits.shape[0], 1, -1).float(), score, tracked, offsets, box, cls, mask_coeff, B, top_k, cols, max_age,
                         score_thr)


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
    "lincomb_sigmoid_crop_bits": _lincomb_bits,
    "mask_iou_bits": _mask_iou_bits,
    "gather_detections": _gather_detections,
    "shift_rois": _shift_rois,
    "shift_apply_": _shift_apply_,
    "match_scores": _match_scores,
    "match_scores_embed": _match_scores_embed,
    "gather_rows2": _gather_rows2,
    "pack_tracked": _pack_tracked,
    "pack_tracked_bits": _pack_tracked_bits,
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
