"""Seeded input of the row-f2 golden (shared by gen_golden.py and tests/test_results_json.py)."""
import torch


def synth_video_results(seed=0):
    """Seeded per-frame tracked detections of two short videos (row f2 input): objects appear / disappear, labels flip
    on some frames (majority vote), one frame has no detections, one object id is -1 (dropped by the reference)."""
    g = torch.Generator().manual_seed(seed)
    frames = []
    for vid, n_frames in ((3, 5), (7, 4)):
        for fid in range(n_frames):
            ids = [i for i in range(4) if torch.rand(1, generator=g).item() < 0.7]
            if vid == 7 and fid == 2:
                ids = []
            n = len(ids)
            obj_ids = torch.tensor(ids, dtype=torch.int64)
            if n and vid == 3 and fid == 1:
                obj_ids[0] = -1
            cls = torch.tensor([(5 + 3 * i + (1 if torch.rand(1, generator=g).item() < 0.25 else 0)) for i in ids], dtype=torch.int64)
            det = {"box": torch.rand(n, 4, generator=g), "class": cls, "score": torch.rand(n, generator=g),
                   "box_ids": obj_ids,
                   "segm": [{"size": [6, 8], "counts": ("%d0%d" % (vid, fid * 10 + i)).encode()} for i in ids]}
            frames.append((det, {"video_id": vid, "frame_id": fid}))
    return frames
