"""Result aggregation, SURVEY.md section 8 row f2: mirrors layers/eval_utils.py:15-106 of the reference.

`bbox2result_with_id` turns one frame's tracked detections (after `output_utils.postprocess_ytbvis`) into the per-object
dict the reference keeps per frame; `results2json_videoseg` folds the ordered per-frame dicts of one or more videos into
YouTube-VIS records: per object the mean score over its frames (float32 mean, as `np.array(scores).mean()` of float32
scalars gives), the majority-vote category (`np.bincount(...).argmax()`), and the per-frame RLE list with None for frames
the object is absent from.  Host-side bookkeeping on the all-gathered detections; no device work.
"""
import json
import os

import numpy as np


def bbox2result_with_id(preds, img_meta, classes):
    """One frame of tracked detections -> the per-frame record `results2json_videoseg` consumes (output contract of
    layers/eval_utils.py:15-50): `{"video_id": v, "frame_id": f, <object id>: {"bbox", "score", "segm"[, "label", "category"]}}`,
    one entry per instance whose tracker id is >= 0 (`remove_false_inst` marks dropped instances with -1).  `preds`: 'box' [n,4],
    'score' [n], 'box_ids' [n], 'class' [n] or None, 'segm' = n RLE dicts; tensors on any device.  Values are numpy scalars /
    rows, exactly as the reference hands them on (numpy float32 scores matter: the per-video mean is taken in float32)."""
    record = {"video_id": img_meta["video_id"], "frame_id": img_meta["frame_id"]}
    n = int(preds["box"].shape[0])
    if n == 0:
        return record
    to_np = lambda t: t.detach().cpu().numpy()
    ids, boxes, scores = to_np(preds["box_ids"]), to_np(preds["box"]), to_np(preds["score"])
    labels = to_np(preds["class"]) if preds["class"] is not None else None
    for row in np.flatnonzero(ids >= 0):
        entry = {"bbox": boxes[row], "score": scores[row], "segm": preds["segm"][row]}
        if labels is not None:
            entry["label"] = labels[row]
            entry["category"] = classes[labels[row] - 1]
        record[ids[row]] = entry
    return record


def video_records(results):
    """The list `results2json_videoseg` (layers/eval_utils.py:53-101) dumps: `results` is the frame-ordered list of
    bbox2result_with_id dicts of one or more videos.  RLE counts given as bytes are decoded to str, as the reference does."""
    json_results = []
    vid_objs = {}
    size = len(results)
    for idx in range(size):
        vid_id, frame_id = results[idx]["video_id"], results[idx]["frame_id"]
        is_last = idx == size - 1 or results[idx + 1]["video_id"] != vid_id
        det = results[idx]
        for obj_id in det:
            if obj_id in ("video_id", "frame_id"):
                continue
            obj = det[obj_id]
            segm = obj["segm"]
            if obj_id not in vid_objs:
                vid_objs[obj_id] = {"scores": [], "cats": [], "segms": {}}
            vid_objs[obj_id]["scores"].append(obj["score"])
            vid_objs[obj_id]["cats"].append(obj["label"])
            if isinstance(segm["counts"], bytes):
                segm["counts"] = segm["counts"].decode()
            vid_objs[obj_id]["segms"][frame_id] = segm
        if is_last:
            for obj_id, obj in vid_objs.items():
                data = {"video_id": vid_id,
                        "score": np.array(obj["scores"]).mean().item(),
                        # majority voting for the sequence category (eval_utils.py:91)
                        "category_id": np.bincount(np.array(obj["cats"])).argmax().item()}
                data["segmentations"] = [obj["segms"].get(fid) for fid in range(frame_id + 1)]
                json_results.append(data)
            vid_objs = {}
    return json_results


def results2json_videoseg(results, out_file):
    """layers/eval_utils.py:53-106: write the records as JSON (mmcv.dump of a .json path is json.dump)."""
    records = video_records(results)
    out_dir = os.path.dirname(out_file)
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir)
    with open(out_file, "w") as f:
        json.dump(records, f)
    return records
