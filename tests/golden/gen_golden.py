#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE's own Python (build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) never travels to the GPU box; only the small .npz files
written here do.  They hold inputs + the outputs of the reference's torch code for the
hot-path functions:

  postproc.npz   decode / center_size / jaccard / sanitize_coordinates_hw / crop /
                 generate_candidate / Detect_TF.detect (cross-class and per-class) /
                 generate_mask / mask_iou on seeded synthetic head outputs
  priors.npz     PredictionModule_FC.make_priors for the five 384x640 level sizes
  fcb_ali.npz    FeatureAlign "ali" offset construction for 3x3 / 3x5 / 5x3
  model_*.npz    STMask.forward (eval) over a seeded 3-frame clip at reduced size for the
                 benchmark configs, with the four third-party ops replaced by the CPU
                 oracle (dcn_v2 / mmcv / spatial_correlation_sampler are not installed
                 anywhere -- "parity unpinned" for those, see oracle/stm_oracle.c)

The reference has no CPU path (STMask.py:15, TF_utils.py:105,109), so the stub preamble
below fakes the CUDA-only bits; nothing of the reference is copied.
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import oracle  # noqa: E402
from stmask_amd import synthetic  # noqa: E402


# ----------------------------------------------------------------------------- stubs
def install_stubs():
    sys.path.insert(0, REF)
    collections.Sequence = collections.abc.Sequence
    np.int = int
    np.float = float
    torch.cuda.current_device = lambda: "cpu"  # TF_utils.py:105,109 pass it as device=

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return None

    def stub(name, **kw):
        m = _Any(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    stub("dcn_v2", DCN=oracle.OracleDCN, DCNv2=oracle.OracleDCN)
    mm = stub("mmcv")
    mm.is_str = lambda x: isinstance(x, str)
    mm.is_list_of = lambda a, b: True
    stub("mmcv.ops", DeformConv2d=oracle.OracleDeformConv2d, roi_align=oracle.roi_align)
    stub("mmcv.runner")
    stub("mmcv.parallel", DataContainer=object)
    stub("spatial_correlation_sampler", spatial_correlation_sample=oracle.spatial_correlation_sample)
    for n in ["cocoapi", "cocoapi.PythonAPI", "cocoapi.PythonAPI.pycocotools", "pycocotools", "pycocotools.mask",
              "cv2", "pyximport"]:
        stub(n)
    sys.modules["pyximport"].install = lambda *a, **k: None
    stub("cocoapi.PythonAPI.pycocotools.ytvos", YTVOS=object)
    stub("cocoapi.PythonAPI.pycocotools.ytvoseval", YTVOSeval=object)
    stub("pycocotools.coco", COCO=object)
    stub("pycocotools.cocoeval", COCOeval=object)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    stub("utils.cython_nms", nms=None)
    import matplotlib  # noqa: F401  (reference imports pyplot at module scope)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.ascontiguousarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={sorted(out)}")


# ----------------------------------------------------------------------------- post-processing
def synth_head_outputs(n_priors_hw, seed, n_hot=400):
    """Seeded synthetic head outputs shaped like forward_single's: loc, softmaxed conf, centerness,
    mask_coeff, track; priors come from the reference's own make_priors."""
    g = torch.Generator().manual_seed(seed)
    N = sum(h * w for h, w in n_priors_hw) * 3
    loc = torch.randn(N, 4, generator=g) * torch.tensor([1.0, 1.0, 1.5, 1.5])
    logits = torch.randn(N, 41, generator=g)
    hot = torch.randperm(N, generator=g)[:n_hot]
    cls = torch.randint(1, 41, (n_hot,), generator=g)
    logits[hot, cls] += torch.rand(n_hot, generator=g) * 6 + 2
    conf = torch.softmax(logits, -1)
    centerness = torch.tanh(torch.randn(N, 1, generator=g) + 1.0)
    coeff = torch.randn(N, 32, generator=g)
    track = torch.nn.functional.normalize(torch.randn(N, 128, generator=g), dim=-1)
    return loc, conf, centerness, coeff, track


def gen_postproc():
    from datasets.config import cfg, set_cfg
    set_cfg("STMask_plus_resnet50_config")
    from layers.box_utils import decode, center_size, jaccard, sanitize_coordinates_hw, crop, mask_iou
    from layers.mask_utils import generate_mask
    from layers.functions import Detect_TF, generate_candidate
    from layers.modules import PredictionModule_FC

    # priors from the reference's own Python loop (prediction_head_FC.py:224-247)
    levels = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]  # reduced pyramid: 1023 priors
    pm = PredictionModule_FC.__new__(PredictionModule_FC)
    pm.pred_aspect_ratios = cfg.backbone.pred_aspect_ratios[0]
    pm.pred_scales = cfg.backbone.pred_scales[0]
    priors = torch.cat([pm.make_priors(h, w, "cpu") for h, w in levels], 1)  # [1,N,4]

    out = {}
    for case, (seed, n_hot) in enumerate([(11, 300), (12, 40), (13, 900)]):
        loc, conf, cen, coeff, track = synth_head_outputs(levels, seed, n_hot)
        g = torch.Generator().manual_seed(100 + seed)
        proto = torch.relu(torch.randn(24, 40, 32, generator=g))
        boxes = decode(loc, priors[0])
        preds = {"loc": loc[None], "conf": conf[None], "priors": priors, "mask_coeff": coeff[None],
                 "track": track[None], "centerness": cen[None], "proto": proto[None],
                 "T2S_feat": torch.zeros(1, 1, 2, 2), "fpn_feat": torch.zeros(1, 1, 2, 2)}
        cand = generate_candidate(preds)[0]
        keep_idx = torch.nonzero(conf[:, 1:].max(1)[0] > cfg.eval_conf_thresh).view(-1)
        det = Detect_TF(cfg.num_classes, bkg_label=0, top_k=cfg.nms_top_k, conf_thresh=cfg.nms_conf_thresh,
                        nms_thresh=cfg.nms_thresh)
        cc = det.detect(dict(cand))
        det.use_cross_class_nms = False
        pc = det.detect(dict(cand))
        masks = generate_mask(cand["proto"], cc["mask_coeff"], cc["box"])
        masks_nocrop = generate_mask(cand["proto"], cc["mask_coeff"][:5], None)
        bin_m = masks.gt(0.5).float()
        miou = mask_iou(bin_m[: min(20, len(bin_m))], bin_m)
        p = f"c{case}_"
        out.update({
            p + "loc": loc, p + "conf": conf, p + "centerness": cen, p + "mask_coeff": coeff,
            p + "track": track[:, :8], p + "proto": proto, p + "boxes": boxes,
            p + "center_size": center_size(boxes), p + "keep_idx": keep_idx,
            p + "cand_box": cand["box"], p + "cand_conf": cand["conf"], p + "cand_centerness": cand["centerness"],
            p + "cc_box": cc["box"], p + "cc_class": cc["class"], p + "cc_score": cc["score"],
            p + "cc_mask_coeff": cc["mask_coeff"], p + "cc_centerness": cc["centerness"],
            p + "pc_box": pc["box"], p + "pc_class": pc["class"], p + "pc_score": pc["score"],
            p + "masks": masks, p + "masks_nocrop": masks_nocrop, p + "mask_iou": miou,
            p + "jaccard": jaccard(cand["box"][:64], cand["box"][:96]),
            p + "sanitize_hw": sanitize_coordinates_hw(cc["box"], 24, 40),
        })
    out["priors"] = priors[0]
    out["levels"] = np.array(levels)
    # crop boundary known-answer case (boxes touching 0 / 1, swapped corners)
    kb = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.5, 0.25, 0.25, 0.75], [0.0, 0.0, 0.024, 0.04],
                       [0.98, 0.97, 1.0, 1.0], [0.3, 0.3, 0.3, 0.3]])
    ones = torch.ones(24, 40, 5)
    cm, _ = crop(ones, kb)
    out["crop_boxes"] = kb
    out["crop_mask"] = cm.permute(2, 0, 1).contiguous()
    save("postproc.npz", **out)


def gen_postproc_nontf():
    """Row a18's per-class Fast NMS: the reference's own Detect.fast_nms (detection.py:211-261) -- which, unlike Detect_TF.fast_nms, ranks by the
    raw class confidences WITHOUT centerness -- on the stored inputs of postproc.npz (outputs only go into postproc_nontf.npz)."""
    from datasets.config import cfg, set_cfg
    set_cfg("STMask_plus_resnet50_config")
    from layers.functions import Detect
    z = np.load(os.path.join(HERE, "postproc.npz"))
    out = {}
    for case in range(3):
        p = f"c{case}_"
        conf, boxes = torch.from_numpy(z[p + "conf"]), torch.from_numpy(z[p + "boxes"])
        coeff = torch.from_numpy(z[p + "mask_coeff"])
        track = torch.nn.functional.normalize(torch.randn(conf.shape[0], 8), dim=-1)        # (carried along only)
        keep = conf[:, 1:].max(1)[0] > cfg.nms_conf_thresh                                   # Detect.detect, detection.py:104-108
        det = Detect(cfg.num_classes, bkg_label=0, top_k=cfg.nms_top_k, conf_thresh=cfg.nms_conf_thresh, nms_thresh=cfg.nms_thresh)
        r = det.fast_nms(boxes[keep], coeff[keep], track[keep], conf[keep, 1:].t().contiguous(), det.nms_thresh, det.top_k)
        out.update({p + "pcn_box": r["box"], p + "pcn_class": r["class"], p + "pcn_score": r["score"], p + "pcn_mask_coeff": r["mask_coeff"]})
    save("postproc_nontf.npz", **out)


def gen_priors():
    from datasets.config import cfg, set_cfg
    set_cfg("STMask_plus_resnet50_config")
    from layers.modules import PredictionModule_FC
    pm = PredictionModule_FC.__new__(PredictionModule_FC)
    pm.pred_aspect_ratios = cfg.backbone.pred_aspect_ratios[0]
    pm.pred_scales = cfg.backbone.pred_scales[0]
    out = {}
    for h, w in [(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)]:
        out[f"p_{h}x{w}"] = pm.make_priors(h, w, "cpu")[0]
    save("priors.npz", **out)


def gen_fcb_ali():
    from datasets.config import set_cfg
    set_cfg("STMask_plus_resnet50_ali_config")
    from layers.modules import FeatureAlign
    out = {}
    g = torch.Generator().manual_seed(5)
    loc = torch.randn(2, 4, 6, 10, generator=g)
    out["loc"] = loc
    for kh, kw in [(3, 3), (3, 5), (5, 3)]:
        fa = FeatureAlign(8, 8, kernel_size=(kh, kw), deformable_groups=1, use_pred_offset=False)
        captured = {}
        fa.conv_adaption.forward = lambda x, off, _c=captured: (_c.__setitem__("off", off.clone()), x)[1]
        fa(torch.zeros(2, 8, 6, 10), loc)
        out[f"off_{kh}x{kw}"] = captured["off"]
    save("fcb_ali.npz", **out)


# ----------------------------------------------------------------------------- whole model
def gen_model(cfg_name, tag, hw=(128, 192), n_frames=3, temporal_fusion=True):
    from datasets.config import cfg, set_cfg
    set_cfg(cfg_name)
    cfg.temporal_fusion_module = temporal_fusion  # False: the reference's Detect / Track path (STMask.py:323-327)
    import STMask as stmask_mod
    net = stmask_mod.STMask()
    net.eval()
    sd = synthetic.fill_state_dict(net, seed=0)
    frames = synthetic.synthetic_clip(n_frames, hw[0], hw[1], seed=0)
    out = {"frames_hw": np.array(hw), "n_frames": np.array(n_frames)}
    out["state_keys"] = np.array(sorted(sd.keys()))
    out["state_shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    with torch.no_grad():
        fpn_outs, po = net.forward_single(frames[:1])
        out["f0_loc"] = po["loc"][0]
        out["f0_conf_logits"] = po["conf"][0]
        out["f0_mask_coeff"] = po["mask_coeff"][0]
        out["f0_centerness"] = po["centerness"][0]
        out["f0_track_s"] = po["track"][0][::7]
        out["f0_proto"] = po["proto"][0]
        out["f0_priors"] = po["priors"][0]
        out["f0_P4"] = fpn_outs[1][0, ::16]
        for t in range(n_frames):
            meta = [{"is_first": t == 0, "video_id": 0, "frame_id": t}]
            res = net(frames[t:t + 1], img_meta=meta)[0]["detection"]
            for k in ("box", "score", "class", "box_ids", "mask_coeff", "mask", "centerness"):
                v = res.get(k, torch.zeros(0))
                v = torch.zeros(0) if v is None else v
                out[f"t{t}_{k}"] = v
            print(tag, "frame", t, "n_out", len(res["box"]))
    save(f"model_{tag}.npz", **out)


def gen_model_nontf(cfg_name, tag, hw=(128, 192), n_frames=3):
    """Row a18: the reference's non-TF post-processing, Detect.detect (detection.py:98-137) + Track.track (track.py:56-179).
    Detect.__call__ cannot run (it reads result['bbox_idx'], detection.py:91, a key cc_fast_nms never sets), so its body is
    driven from here exactly as __call__ does it (detection.py:69-90) minus that line: transpose conf, decode, detect(batch_idx,
    ...), result['proto'] = proto; then Track.track(result, meta) per frame as STMask.forward does (STMask.py:323-327)."""
    from datasets.config import cfg, set_cfg
    set_cfg(cfg_name)
    cfg.temporal_fusion_module = False
    import STMask as stmask_mod
    from layers.box_utils import decode
    net = stmask_mod.STMask()
    net.eval()
    synthetic.fill_state_dict(net, seed=0)
    frames = synthetic.synthetic_clip(n_frames, hw[0], hw[1], seed=0)
    out = {"frames_hw": np.array(hw), "n_frames": np.array(n_frames)}
    with torch.no_grad():
        for t in range(n_frames):
            _, po = net.forward_single(frames[t:t + 1])
            po["conf"] = torch.softmax(po["conf"], -1)
            po["mask_coeff"] = cfg.mask_proto_coeff_activation(po["mask_coeff"])      # STMask.py:324
            pri = po["priors"].squeeze(0)
            conf_t = po["conf"].view(1, pri.size(0), -1).transpose(2, 1).contiguous()   # detection.py:77-78
            boxes = decode(po["loc"][0], pri)
            res = net.detect.detect(0, conf_t, boxes, po["centerness"], po["mask_coeff"], po["track"], po["proto"], None)
            res["proto"] = po["proto"][0]
            out[f"t{t}_nms_box"], out[f"t{t}_nms_class"], out[f"t{t}_nms_score"] = res["box"], res["class"], res["score"]
            res = net.Track.track(res, {"is_first": t == 0, "video_id": 0, "frame_id": t})
            for k in ("box", "score", "class", "box_ids", "mask_coeff", "mask"):
                v = res.get(k, torch.zeros(0))
                out[f"t{t}_{k}"] = torch.zeros(0) if v is None else v
            print(tag, "frame", t, "n_nms", len(out[f"t{t}_nms_box"]), "n_out", len(res["box"]))
    save(f"model_{tag}.npz", **out)


def gen_model_full(cfg_name, tag, hw=(384, 640), row_step=16):
    """ONE full-size frame through the reference's STMask.forward (eval, oracle ops plugged in) at the benchmark's
    own weights (synthetic.BENCH_BG_BIAS).  Stored: float64 checksums (sum, sum |.|) of every head output, strided slices of
    them, and the frame's detections (boxes / classes / scores / soft masks)."""
    from datasets.config import cfg, set_cfg
    set_cfg(cfg_name)
    cfg.temporal_fusion_module = True
    import STMask as stmask_mod
    net = stmask_mod.STMask()
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    frame = synthetic.synthetic_clip(1, hw[0], hw[1], seed=0)
    out = {"frames_hw": np.array(hw), "row_step": np.array(row_step)}
    with torch.no_grad():
        fpn_outs, po = net.forward_single(frame)
        full = {"loc": po["loc"][0], "conf_logits": po["conf"][0], "mask_coeff": po["mask_coeff"][0],
                "centerness": po["centerness"][0], "track": po["track"][0], "proto": po["proto"][0], "P4": fpn_outs[1][0]}
        for k, v in full.items():
            out[f"sum_{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), float(v.abs().max())])
        for k in ("loc", "conf_logits", "mask_coeff", "centerness"):
            out[f"s_{k}"] = full[k][::row_step]
        out["s_track"] = full["track"][::row_step, ::8]
        out["s_proto"] = full["proto"][::4, ::4]
        out["s_P4"] = full["P4"][::16]
        out["n_priors"] = np.array(full["loc"].shape[0])
        res = net(frame, img_meta=[{"is_first": True, "video_id": 0, "frame_id": 0}])[0]["detection"]
        n = min(len(res["box"]), 48)
        for k in ("box", "score", "class", "box_ids", "mask_coeff", "mask", "centerness"):
            out[f"det_{k}"] = res[k][:n]
        out["det_n"] = np.array(len(res["box"]))
        print(tag, "full-size frame", hw, "priors", full["loc"].shape[0], "detections", len(res["box"]))
    save(f"model_full_{tag}.npz", **out)


FRAGILE_EPS = 1e-4


def gen_model_full_tf(cfg_name, tag, hw=(384, 640), n_frames=3, n_masks=32):
    """frames 0..2 of a FULL-SIZE clip through the reference's eval forward (STMask.py:310-329) with the benchmark's weights:
    frame 0 detects, frames 1-2 run CandidateShift (TF_utils.py:12-51: correlation, RoIAlign, TemporalNet, decode, lincomb on the current
    prototypes) and Track_TF.track (track_TF.py:50-181: comp scores, greedy resolution, ids, keep rule) on ~100 tracked instances.  Stored per
    frame: every tracked instance's box / score / class / id / coefficients / centerness, float64 checksums (sum, sum of squares, count of pixels
    > 0.5) of EVERY soft mask, the first n_masks soft masks whole, and checksums of the frame's head outputs."""
    from datasets.config import cfg, set_cfg
    set_cfg(cfg_name)
    cfg.temporal_fusion_module = True
    import STMask as stmask_mod
    net = stmask_mod.STMask()
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    frames = synthetic.synthetic_clip(n_frames, hw[0], hw[1], seed=0)
    out = {"frames_hw": np.array(hw), "n_frames": np.array(n_frames), "n_masks": np.array(n_masks), "fragile_eps": np.array(FRAGILE_EPS)}
    # Which rows of the tracker state hang on a near-tie.  Track_TF.track decides by argmax over comp scores (track_TF.py:125) and, among detections that
    # pick the same object, by det_score > best (track_TF.py:141); with 120-200 overlapping instances some of those comparisons are closer than the
    # ~1e-6 by which two correct fp32 implementations of the trunk differ.  Recorded from the reference's OWN values, per frame: the rows whose outcome
    # a perturbation below FRAGILE_EPS could change (the test compares all other rows exactly) and whether the row COUNT could change.
    import layers.functions.track_TF as ttf
    seen = {}
    real_ccs = ttf.compute_comp_scores

    def spy(match_ll, bbox_scores, *a, **k):
        comp = real_ccs(match_ll, bbox_scores, *a, **k)
        seen["comp"], seen["score"] = comp.clone(), bbox_scores.view(-1).clone()
        return comp
    ttf.compute_comp_scores = spy
    with torch.no_grad():
        for t in range(n_frames):
            seen.clear()
            _, po = net.forward_single(frames[t:t + 1])
            for k in ("loc", "conf", "mask_coeff", "centerness", "proto"):
                v = po[k][0].double()
                out[f"t{t}_sum_{k}"] = np.array([v.sum().item(), v.abs().sum().item()])
            res = net(frames[t:t + 1], img_meta=[{"is_first": t == 0, "video_id": 0, "frame_id": t}])[0]["detection"]
            for k in ("box", "score", "class", "box_ids", "mask_coeff", "centerness"):
                out[f"t{t}_{k}"] = res[k]
            m = res["mask"].double()
            out[f"t{t}_mask_sums"] = torch.stack([m.sum(dim=(1, 2)), m.pow(2).sum(dim=(1, 2)), (m > 0.5).double().sum(dim=(1, 2))], 1)
            out[f"t{t}_mask"] = res["mask"][:n_masks]
            # the tracker's whole state after the frame (row = instance id): what CandidateShift moved and Track_TF.track matched / appended,
            # including the instances the keep rule (track_TF.py:168-178) holds back from the output
            st = net.Track_TF.prev_candidate
            for k in ("box", "score", "class", "mask_coeff", "tracked_mask"):
                out[f"t{t}_state_{k}"] = st[k].clone()          # (the next frame updates matched rows IN PLACE: track_TF.py:150)
            ms = st["mask"].double()
            out[f"t{t}_state_mask_sums"] = torch.stack([ms.sum(dim=(1, 2)), ms.pow(2).sum(dim=(1, 2)), (ms > 0.5).double().sum(dim=(1, 2))], 1)
            fragile, count_fragile, min_margin = set(), False, float("inf")
            if "comp" in seen:
                comp, dsc = seen["comp"], seen["score"]
                n_prev = comp.shape[1] - 1
                top2v, top2i = comp.topk(2, dim=1)
                cm = top2v[:, 0] - top2v[:, 1]
                mid = top2i[:, 0]
                new_rank = torch.cumsum((mid == 0).long(), 0) - 1                      # appended row of a new object = n_prev + rank
                for i in range(comp.shape[0]):
                    if cm[i] < FRAGILE_EPS:
                        for j in top2i[i].tolist():
                            if j == 0:
                                count_fragile = True
                            else:
                                fragile.add(j - 1)
                        if mid[i] == 0:
                            fragile.add(n_prev + int(new_rank[i]))
                for obj in range(n_prev):
                    c = torch.nonzero(mid == obj + 1).view(-1)
                    if len(c) > 1:
                        sc = dsc[c].sort(descending=True).values
                        if sc[0] - sc[1] < FRAGILE_EPS:
                            fragile.add(obj)
                        min_margin = min(min_margin, float(sc[0] - sc[1]))
                min_margin = min(min_margin, float(cm.min()))
            out[f"t{t}_fragile_rows"] = np.array(sorted(fragile), dtype=np.int64)
            out[f"t{t}_count_fragile"] = np.array(count_fragile)
            out[f"t{t}_min_margin"] = np.array(min_margin)
            print(tag, "state rows", len(st["box"]), "fragile rows", sorted(fragile), "count fragile", count_fragile, "min margin %.2e" % min_margin)
            print(tag, "full-size TF frame", t, "tracked", len(res["box"]), "ids", int(res["box_ids"].max()) + 1 if len(res["box"]) else 0)
    ttf.compute_comp_scores = real_ccs
    save(f"model_full_tf_{tag}.npz", **out)


from synth_results import synth_video_results  # noqa: E402  (tests/golden/synth_results.py, shared with the test)


def gen_results_json():
    """Row f2: the reference's own bbox2result_with_id + results2json_videoseg (layers/eval_utils.py) on seeded inputs."""
    import json
    sys.modules["mmcv"].dump = lambda obj, path: json.dump(obj, open(path, "w"))
    from layers import eval_utils
    classes = ["c%d" % i for i in range(40)]
    frames = synth_video_results()
    results = [eval_utils.bbox2result_with_id(det, meta, classes) for det, meta in frames]
    per_frame = [{str(k): (v if k in ("video_id", "frame_id") else
                           {"bbox": v["bbox"].tolist(), "label": int(v["label"]), "score": float(v["score"]),
                            "category": v["category"]}) for k, v in r.items()} for r in results]
    out = os.path.join(HERE, "_tmp_results_dir", "results.json")   # the reference makedirs(out_file[:-13])
    os.makedirs(os.path.dirname(out), exist_ok=True)
    eval_utils.results2json_videoseg(results, out)
    records = json.load(open(out))
    os.remove(out)
    os.rmdir(os.path.dirname(out))
    with open(os.path.join(HERE, "results_json.json"), "w") as f:
        json.dump({"per_frame": per_frame, "records": records}, f)
    print("wrote results_json.json:", len(records), "records")


def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["postproc", "priors", "fcb_ali", "model"]
    if "postproc" in which:
        gen_postproc()
    if "postproc_nontf" in which:
        gen_postproc_nontf()
    if "priors" in which:
        gen_priors()
    if "fcb_ali" in which:
        gen_fcb_ali()
    if "model" in which:
        gen_model("STMask_plus_resnet50_config", "r50_fca")
        gen_model("STMask_plus_resnet50_ada_config", "r50_ada")
        gen_model("STMask_plus_resnet50_ali_config", "r50_ali")
    if "results_json" in which:
        gen_results_json()
    if "model_nontf" in which:
        gen_model_nontf("STMask_plus_resnet50_config", "r50_fca_nontf")
    if "model_full" in which:
        gen_model_full("STMask_plus_resnet50_config", "r50_fca")
        gen_model_full("STMask_plus_resnet50_ada_config", "r50_ada")
        gen_model_full("STMask_plus_base_ali_config", "r101_ali")
    if "model_full_tf" in which:
        gen_model_full_tf("STMask_plus_resnet50_config", "r50_fca")
        gen_model_full_tf("STMask_plus_resnet50_ada_config", "r50_ada")
    if "model_full_720p" in which:
        gen_model_full("STMask_plus_base_ali_config", "r101_ali_736x1280", hw=(736, 1280), row_step=64)
    if "model_extra" in which:
        gen_model("STMask_plus_base_ali_config", "r101_ali", hw=(96, 160), n_frames=2)


if __name__ == "__main__":
    main()
