"""Post-processing functions of the hot path -- host-side mirror of the reference's layers/functions/{TF_utils,
detection_TF,track_TF,detection,track}.py: same callables, same dict-of-tensors interchange
(box, score, class, mask_coeff, track, centerness, proto, fpn_feat, T2S_feat, mask, box_ids).

Tensor math runs in the hand-written gfx950 kernels (decode + threshold + compaction, Fast NMS, lincomb + crop,
correlation, RoIAlign, box / mask IoU).  The stateful tracker keeps the reference's data-dependent control flow on the
host, but reads back ONE small tensor per frame (the match ids) instead of syncing per detection
(reference track_TF.py:132-156).
"""
import torch
import torch.nn.functional as F

from .. import ops
from ..config import cfg as _default_cfg
from .box_utils import center_size, decode, jaccard, mask_iou
from .mask_utils import generate_mask
from .modules import bbox_feat_extractor, correlate

_FEATURE_KEYS = ("proto", "fpn_feat", "T2S_feat")
ROI_BUCKET = 32


def _empty(dev):
    return torch.zeros(0, device=dev)


# ------------------------------------------------------------------------------------------ candidates
def generate_candidate(predictions, cfg=None):
    """Reference TF_utils.py:54-82: decode, keep rows with max foreground confidence > eval_conf_thresh, gather.
    One fused launch produces decoded boxes + ordered keep indices for the whole batch; the remaining per-key gathers
    are torch index_selects (the only host sync is reading the per-frame counts, as in the reference)."""
    cfg = cfg or _default_cfg
    loc, conf = predictions["loc"], predictions["conf"]
    priors = predictions["priors"].squeeze(0)
    keep_idx, cand_box, count = ops.generate_candidates(loc, priors, conf, cfg.eval_conf_thresh)
    counts = ops.counts_to_host(count)   # + the fp16 range flag of a fp16x2 planar graph
    out = []
    for i, k in enumerate(counts):
        idx = keep_idx[i, :k]
        cur = {"T2S_feat": predictions["T2S_feat"][i].unsqueeze(0), "fpn_feat": predictions["fpn_feat"][i].unsqueeze(0),
               "proto": predictions["proto"][i], "conf": conf[i].index_select(0, idx), "box": cand_box[i, :k],
               "mask_coeff": predictions["mask_coeff"][i].index_select(0, idx),
               "track": predictions["track"][i].index_select(0, idx) if cfg.train_track else None}
        if cfg.train_centerness:
            cur["centerness"] = predictions["centerness"][i].index_select(0, idx).view(-1)
        out.append(cur)
    return out


def merge_candidates(candidate, ref_candidate_clip_shift):
    """Reference TF_utils.py:85-96."""
    merged = {k: v.clone() for k, v in candidate.items()}
    for ref in ref_candidate_clip_shift:
        if ref["box"].nelement() > 0:
            for k, v in merged.items():
                if k not in _FEATURE_KEYS:
                    merged[k] = torch.cat([v, ref[k]], dim=0)
    return merged


def compute_comp_scores(match_ll, bbox_scores, bbox_ious, mask_ious, label_delta, add_bbox_dummy=False,
                        bbox_dummy_iou=0, match_coeff=None):
    """Reference TF_utils.py:99-120 (operand order preserved)."""
    if add_bbox_dummy:
        dummy = torch.ones(bbox_ious.size(0), 1, device=bbox_ious.device) * bbox_dummy_iou
        bbox_ious = torch.cat((dummy, bbox_ious), dim=1)
        mask_ious = torch.cat((dummy, mask_ious), dim=1)
        label_delta = torch.cat((torch.ones_like(dummy), label_delta), dim=1)
    if match_coeff is None:
        return match_ll
    assert len(match_coeff) == 4
    return match_ll + match_coeff[0] * bbox_scores + match_coeff[1] * mask_ious + match_coeff[2] * bbox_ious \
        + match_coeff[3] * label_delta


def CandidateShift(net, ref_candidate, next_candidate, img=None, img_meta=None, display=False, cfg=None):
    """Reference TF_utils.py:12-51: move the previous frame's instances onto the current frame.
    correlation(P4_prev, P4_cur) ++ T2S_prev ++ T2S_cur -> ReLU -> RoIAlign 7x7 on the previous boxes -> TemporalNet
    -> (dbox, dcoeff) -> decode / add -> masks on the CURRENT prototypes."""
    cfg = cfg or _default_cfg
    shifted = {k: v.clone() for k, v in next_candidate.items() if k in _FEATURE_KEYS}
    x_corr = correlate(ref_candidate["fpn_feat"], next_candidate["fpn_feat"], patch_size=cfg.correlation_patch_size)
    feats = F.relu(torch.cat([x_corr, ref_candidate["T2S_feat"], next_candidate["T2S_feat"]], dim=1))
    box_ref = ref_candidate["box"].clone()
    feat_h, feat_w = ref_candidate["fpn_feat"].shape[2:]
    roi_feats = bbox_feat_extractor(feats, box_ref, feat_h, feat_w, 7)
    n = roi_feats.shape[0]
    ptn = getattr(net, "_planar_temporal", None)    # fuse.optimize_for_inference(net, planar=True)
    if ptn is not None and n > 0:
        loc_shift, coeff_shift = ptn(roi_feats)
    else:
        n_pad = -(-n // ROI_BUCKET) * ROI_BUCKET
        if n_pad != n:
            roi_feats = torch.cat([roi_feats, roi_feats.new_zeros(n_pad - n, *roi_feats.shape[1:])], 0)
        # fixed-shape blocks: the dense-conv library (MIOpen) selects / builds kernels per shape and the tracked set
        # changes size every frame; rows are independent, so zero padding + blocking is exact
        outs = [net.TemporalNet(roi_feats[i:i + ROI_BUCKET]) for i in range(0, n_pad, ROI_BUCKET)]
        loc_shift = torch.cat([o[0] for o in outs], 0)[:n]
        coeff_shift = torch.cat([o[1] for o in outs], 0)[:n]
    box_shift = decode(loc_shift, center_size(box_ref))
    coeff = ref_candidate["mask_coeff"] + coeff_shift
    shifted["box"] = box_shift
    shifted["score"] = ref_candidate["score"] * 0.95
    shifted["mask_coeff"] = coeff
    shifted["mask"] = generate_mask(next_candidate["proto"], coeff, box_shift)
    return shifted


# ------------------------------------------------------------------------------------------ detection
class Detect_TF(object):
    """Reference detection_TF.py:8-204.  `detect` works on one candidate dict; the NMS itself is one kernel launch."""

    def __init__(self, num_classes, bkg_label, top_k, conf_thresh, nms_thresh, cfg=None):
        if nms_thresh <= 0:
            raise ValueError("nms_threshold must be non negative.")
        self.num_classes, self.background_label, self.top_k = num_classes, bkg_label, top_k
        self.nms_thresh, self.conf_thresh = nms_thresh, conf_thresh
        self.use_cross_class_nms = True
        self.use_fast_nms = True
        # per-class Fast NMS ranks by conf * centerness in Detect_TF (detection_TF.py:139-141) but by the raw class confidences in the non-TF
        # Detect (detection.py:211-212, no centerness argument, no 'centerness' key in its output): Detect clears this on its inner detector
        self.pc_centerness = True
        self.cfg = cfg or _default_cfg

    def __call__(self, net, candidates, is_output_candidate=False):
        results = []
        for candidate in candidates:
            result = self.detect(candidate, is_output_candidate)
            results.append(result if is_output_candidate else {"detection": result, "net": net})
        return results

    def detect(self, candidate, is_output_candidate=False):
        boxes = candidate["box"]
        if boxes.size(0) == 0:
            dev = boxes.device
            out = {"box": boxes, "mask_coeff": candidate["mask_coeff"], "class": _empty(dev), "score": _empty(dev)}
        elif self.use_cross_class_nms:
            out = self.cc_fast_nms(boxes, candidate["mask_coeff"], candidate["proto"], candidate["track"],
                                   candidate["conf"], candidate["centerness"], self.nms_thresh, self.top_k)
        else:
            out = self.fast_nms(boxes, candidate["mask_coeff"], candidate["proto"], candidate["track"],
                                candidate["conf"], candidate["centerness"], self.nms_thresh, self.top_k)
        if is_output_candidate:
            for k, v in candidate.items():
                if k in ("fpn_feat", "proto", "T2S_feat", "sem_seg"):
                    out[k] = v
        return out

    @staticmethod
    def _gather(idx, cls, score, boxes, masks_coeff, track, centerness):
        return {"box": boxes.index_select(0, idx), "mask_coeff": masks_coeff.index_select(0, idx),
                "track": track.index_select(0, idx) if track is not None else None, "class": cls, "score": score,
                "centerness": centerness.index_select(0, idx) if centerness is not None else None}

    def cc_fast_nms(self, boxes, masks_coeff, proto_data, track, conf, centerness_scores, iou_threshold=0.5,
                    top_k=200):
        """conf is the candidate rows [K, num_classes] (the reference passes its transpose without background)."""
        if self.cfg.nms_as_miou:
            raise NotImplementedError("nms_as_miou is False in every STMask config (config.py:721)")
        idx, cls, score, _, count = ops.cc_fast_nms(conf, boxes, centerness_scores, iou_threshold, top_k)
        n = int(count)
        return self._gather(idx[:n], cls[:n], score[:n], boxes, masks_coeff, track, centerness_scores)

    def fast_nms(self, boxes, masks_coeff, proto_data, track, conf, centerness_scores, iou_threshold=0.5, top_k=200,
                 second_threshold=True):
        if not second_threshold:
            raise NotImplementedError("the reference always applies the second threshold")
        if not self.pc_centerness:
            centerness_scores = None
        idx, cls, score, _, count = ops.fast_nms(conf, boxes, centerness_scores, iou_threshold, top_k, self.conf_thresh,
                                                 self.cfg.max_num_detections)
        n = int(count)
        out = self._gather(idx[:n], cls[:n], score[:n], boxes, masks_coeff, track, centerness_scores)
        if out["centerness"] is not None:
            out["centerness"] = out["centerness"].view(-1, 1)  # the reference keeps [n,1] on this path (:152,:201)
        return out


# ------------------------------------------------------------------------------------------ tracking
class Track_TF(object):
    """Reference track_TF.py:16-181: stateful per-video tracker with temporal fusion (batch size 1 per tracker)."""

    _ROW_KEYS_EXCLUDE = ("proto", "T2S_feat", "fpn_feat", "tracked_mask")

    def __init__(self, cfg=None):
        self.prev_candidate = None
        self.cfg = cfg or _default_cfg

    def __call__(self, net, candidates, imgs_meta, imgs=None):
        results = []
        for b, candidate in enumerate(candidates):
            det = self.track(net, candidate, imgs_meta[b], img=None if imgs is None else imgs[b])
            results.append({"detection": det, "net": net})
        return results

    def _shift_prev(self, net, candidate, img, img_meta):
        shifted = CandidateShift(net, self.prev_candidate, candidate, img=img, img_meta=img_meta, cfg=self.cfg)
        self.prev_candidate.update(shifted)
        self.prev_candidate["tracked_mask"] = self.prev_candidate["tracked_mask"] + 1

    def track(self, net, candidate, img_meta, img=None):
        cfg = self.cfg
        dev = candidate["proto"].device if "proto" in candidate else candidate["box"].device
        if img_meta["is_first"]:
            self.prev_candidate = None
        no_dets = candidate["box"].nelement() == 0
        if no_dets and self.prev_candidate is None:
            return {"box": _empty(dev), "mask_coeff": _empty(dev), "class": _empty(dev), "score": _empty(dev),
                    "box_ids": _empty(dev)}
        if no_dets:
            self._shift_prev(net, candidate, img, img_meta)
        else:
            det_bbox, det_score, det_labels = candidate["box"], candidate["score"], candidate["class"]
            det_track = candidate["track"] if cfg.train_track else F.normalize(candidate["mask_coeff"], dim=1)
            n_dets = det_bbox.size(0)
            det_masks_soft = generate_mask(candidate["proto"], candidate["mask_coeff"], det_bbox)
            candidate["mask"] = det_masks_soft
            if self.prev_candidate is None:
                self.prev_candidate = dict(candidate)
                self.prev_candidate["tracked_mask"] = torch.zeros(n_dets)  # host-side counter
            else:
                self._shift_prev(net, candidate, img, img_meta)
                prev = self.prev_candidate
                n_prev = prev["box"].size(0)
                cos_sim = det_track @ prev["track"].t()
                cos_sim = torch.cat([cos_sim.new_zeros(n_dets, 1), cos_sim], dim=1)
                cos_sim = (cos_sim + 1) / 2
                bbox_ious = jaccard(det_bbox, prev["box"])
                mask_ious = mask_iou(det_masks_soft, prev["mask"])  # both sides binarised with > 0.5 in the kernel
                label_delta = (prev["class"] == det_labels.view(-1, 1)).float()
                comp = compute_comp_scores(cos_sim, det_score.view(-1, 1), bbox_ious, mask_ious, label_delta,
                                           add_bbox_dummy=True, bbox_dummy_iou=0.3, match_coeff=cfg.match_coeff)
                match_ids = comp.argmax(dim=1)
                # ONE device->host transfer per frame; the greedy resolution below is the reference's loop
                # (track_TF.py:132-156) on host scalars
                host = torch.stack([match_ids.float(), det_score]).cpu()
                ids, scores = host[0].long().tolist(), host[1].tolist()
                src = list(range(n_prev))          # row of cat([prev, cand]) that ends up in each slot
                best_score = [-1.0] * n_prev
                reset = []                         # slots whose tracked_mask becomes 0
                for idx, mid in enumerate(ids):
                    if mid == 0:
                        src.append(n_prev + idx)
                    else:
                        obj = mid - 1
                        if scores[idx] > best_score[obj]:
                            best_score[obj] = scores[idx]
                            src[obj] = n_prev + idx
                            reset.append(obj)
                plan = torch.tensor(src, device=dev, dtype=torch.int64)
                for k, v in list(prev.items()):
                    if k not in self._ROW_KEYS_EXCLUDE:
                        prev[k] = torch.cat([v, candidate[k]], dim=0).index_select(0, plan)
                tm = torch.cat([prev["tracked_mask"], torch.zeros(len(src) - n_prev)])
                if reset:
                    tm[torch.tensor(reset, dtype=torch.int64)] = 0
                prev["tracked_mask"] = tm

        prev = self.prev_candidate
        n_obj = prev["box"].size(0)
        cond1 = (prev["tracked_mask"] <= 10).to(dev)
        cond2 = prev["mask"].gt(0.5).sum([1, 2]) > 1
        cond3 = prev["score"] > cfg.eval_conf_thresh
        keep = cond1 & cond2 & cond3
        kidx = torch.nonzero(keep).view(-1)  # sync: the output size is data dependent (as in the reference)
        if kidx.numel() == 0:
            return {"box": _empty(dev), "mask_coeff": _empty(dev), "class": _empty(dev), "score": _empty(dev),
                    "box_ids": _empty(dev)}
        det = {k: prev[k].index_select(0, kidx) for k in ("box", "mask_coeff", "track", "class", "score", "centerness",
                                                          "mask")}
        det["proto"] = candidate["proto"]
        det["box_ids"] = torch.arange(n_obj, device=dev).index_select(0, kidx)
        return det


class Detect(object):
    """Non-TF detector (reference detection.py:15-263): same decode / threshold / Fast NMS on raw head outputs."""

    def __init__(self, num_classes, bkg_label, top_k, conf_thresh, nms_thresh, cfg=None):
        if nms_thresh <= 0:
            raise ValueError("nms_threshold must be non negative.")
        self.num_classes, self.background_label, self.top_k = num_classes, bkg_label, top_k
        self.nms_thresh, self.conf_thresh = nms_thresh, conf_thresh
        self.use_cross_class_nms = True
        self.use_fast_nms = True
        self.cfg = cfg or _default_cfg
        self._tf = Detect_TF(num_classes, bkg_label, top_k, conf_thresh, nms_thresh, cfg=self.cfg)
        self._tf.pc_centerness = False       # detection.py:130: fast_nms(boxes, masks_coeff, track, scores, ...) -- centerness only reaches cc_fast_nms

    def __call__(self, predictions, net):
        cfg = self.cfg
        priors = predictions["priors"].squeeze(0)
        keep_idx, cand_box, count = ops.generate_candidates(predictions["loc"], priors, predictions["conf"],
                                                            self.conf_thresh)
        out = []
        for b, k in enumerate(count.tolist()):
            idx = keep_idx[b, :k]
            dev = idx.device
            if k == 0:
                result = {"box": cand_box[b, :0], "mask_coeff": predictions["mask_coeff"][b][:0], "class": _empty(dev),
                          "score": _empty(dev), "bbox_idx": _empty(dev)}
            else:
                self._tf.use_cross_class_nms = self.use_cross_class_nms
                cand = {"box": cand_box[b, :k], "conf": predictions["conf"][b].index_select(0, idx),
                        "mask_coeff": predictions["mask_coeff"][b].index_select(0, idx),
                        "track": predictions["track"][b].index_select(0, idx) if cfg.train_track else None,
                        "centerness": predictions["centerness"][b].index_select(0, idx).view(-1)
                        if cfg.train_centerness else None, "proto": predictions["proto"][b]}
                result = self._tf.detect(cand)
            result["proto"] = predictions["proto"][b]
            out.append({"detection": result, "net": net})
        return out


class Track(object):
    """Non-TF tracker (reference track.py:16-179): keeps binary masks, no temporal fusion."""

    def __init__(self, cfg=None):
        self.cfg = cfg or _default_cfg
        self.prev = None

    def __call__(self, pred_outs_after_NMS, img_meta):
        for b, item in enumerate(pred_outs_after_NMS):
            item["detection"] = self.track(item["detection"], img_meta[b])
        return pred_outs_after_NMS

    def track(self, detection, img_meta):
        cfg = self.cfg
        if img_meta["is_first"]:
            self.prev = None
        dev = detection["box"].device
        if detection["class"].nelement() == 0:
            detection["box_ids"] = torch.zeros(0, dtype=torch.int64, device=dev)
            return detection
        det_bbox, det_labels, det_score = detection["box"], detection["class"], detection["score"]
        det_coeff, proto = detection["mask_coeff"], detection["proto"]
        det_track = detection["track"] if cfg.train_track else F.normalize(det_coeff, dim=1)
        n_dets = det_bbox.size(0)
        # NB: on this path STMask.forward has already applied tanh to mask_coeff (STMask.py:324) and generate_mask
        # applies it again (mask_utils.py:112) -- reproduced as is
        det_masks = generate_mask(proto, det_coeff, det_bbox).gt(0.5).float()
        detection["mask"] = det_masks
        if self.prev is None:
            det_obj_ids = torch.arange(n_dets, device=dev)
            self.prev = {"box": det_bbox, "track": det_track, "class": det_labels.view(-1), "mask": det_masks,
                         "mask_coeff": det_coeff, "score": det_score}
        else:
            p = self.prev
            n_prev = p["box"].size(0)
            cos_sim = det_track @ p["track"].t()
            cos_sim = (torch.cat([cos_sim.new_zeros(n_dets, 1), cos_sim], dim=1) + 1) / 2
            bbox_ious = jaccard(det_bbox, p["box"])
            mask_ious = mask_iou(det_masks, p["mask"])
            label_delta = (p["class"] == det_labels.view(-1, 1)).float()
            comp = compute_comp_scores(cos_sim, det_score.view(-1, 1), bbox_ious, mask_ious, label_delta,
                                       add_bbox_dummy=True, bbox_dummy_iou=0.3, match_coeff=cfg.match_coeff)
            host = torch.stack([comp.argmax(dim=1).float(), det_score, (mask_ious > 0.3).sum(1).float()]).cpu()
            ids, scores, n_over = host[0].long().tolist(), host[1].tolist(), host[2].tolist()
            obj_ids = [-1] * n_dets
            src = list(range(n_prev))
            best_score, best_idx = [-1.0] * n_prev, [-1] * n_prev
            for idx, mid in enumerate(ids):
                if mid == 0:
                    obj_ids[idx] = len(src)
                    src.append(n_prev + idx)
                else:
                    obj = mid - 1
                    if scores[idx] > best_score[obj]:
                        if best_idx[obj] != -1:
                            obj_ids[best_idx[obj]] = -1
                        obj_ids[idx] = obj
                        best_score[obj], best_idx[obj] = scores[idx], idx
                        if n_over[idx] < 2:  # track.py:162
                            src[obj] = n_prev + idx
            plan = torch.tensor(src, device=dev, dtype=torch.int64)
            cand = {"box": det_bbox, "track": det_track, "class": det_labels.view(-1), "mask": det_masks,
                    "mask_coeff": det_coeff, "score": det_score}
            for k in p:
                p[k] = torch.cat([p[k], cand[k]], dim=0).index_select(0, plan)
            det_obj_ids = torch.tensor(obj_ids, device=dev, dtype=torch.int64)
        detection["box_ids"] = det_obj_ids
        if cfg.remove_false_inst:
            keep = torch.nonzero(det_obj_ids >= 0).view(-1)
            for k, v in list(detection.items()):
                if k not in ("proto", "bbox_idx", "priors", "loc_t") and v is not None:
                    detection[k] = v.index_select(0, keep)
        return detection
