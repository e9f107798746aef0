"""Result aggregation, SURVEY.md section 8 row f2: mirrors layers/eval_utils.py:15-106 of the reference.

`bbox2result_with_id` turns one frame's tracked detections (after `output_utils.postprocess_ytbvis`) into the per-object
dict the reference keeps per frame; `results2json_videoseg` folds the ordered per-frame dicts of one or more videos into
YouTube-VIS records: per object the mean score over its frames (float32 mean, as `np.array(scores).mean()` of float32
scalars gives), the majority-vote category (`np.bincount(...).argmax()`), and the per-frame RLE list with None for frames
the object is absent from.  Host-side bookkeeping on the all-gathered detections; no device work.
"""
import itertools
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


_META_KEYS = ("video_id", "frame_id")


class _ObjectTrack:
    """One tracked object of one video: what the YouTube-VIS record needs from its per-frame entries."""

    __slots__ = ("scores", "labels", "rle_by_frame")

    def __init__(self):
        self.scores, self.labels, self.rle_by_frame = [], [], {}

    def add(self, frame_id, entry):
        rle = entry["segm"]
        if isinstance(rle["counts"], bytes):          # pycocotools hands the counts out as bytes; JSON wants str
            rle["counts"] = rle["counts"].decode()
        self.scores.append(entry["score"])
        self.labels.append(entry["label"])
        self.rle_by_frame[frame_id] = rle

    def record(self, video_id, n_frames):
        return {"video_id": video_id,
                "score": np.array(self.scores).mean().item(),                       # float32 mean of float32 scalars
                "category_id": np.bincount(np.array(self.labels)).argmax().item(),  # majority vote, lowest id wins a tie
                "segmentations": [self.rle_by_frame.get(f) for f in range(n_frames)]}


def video_records(results):
    """Output contract of `results2json_videoseg` (layers/eval_utils.py:53-101): `results` is the frame-ordered list of
    bbox2result_with_id dicts of one or more videos (a video = a run of consecutive entries with one video_id); the return
    value is one record per (video, object), videos in order of appearance and objects in order of first appearance, each
    with the object's mean score, its majority category and one RLE (or None) per frame up to the video's last frame id."""
    records = []
    for video_id, run in itertools.groupby(results, key=lambda r: r["video_id"]):
        tracks, last_frame = {}, -1
        for frame in run:
            last_frame = frame["frame_id"]
            for obj_id, entry in frame.items():
                if obj_id not in _META_KEYS:
                    tracks.setdefault(obj_id, _ObjectTrack()).add(last_frame, entry)
        records.extend(t.record(video_id, last_frame + 1) for t in tracks.values())
    return records


def results2json_videoseg(results, out_file):
    """layers/eval_utils.py:53-106: write the records as JSON (mmcv.dump of a .json path is json.dump)."""
    records = video_records(results)
    out_dir = os.path.dirname(out_file)
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir)
    with open(out_file, "w") as f:
        json.dump(records, f)
    return records
