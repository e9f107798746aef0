#!/usr/bin/env python3
"""End-to-end flow on the GPU, the way eval.py drives the reference, with every stage on this repository's kernels:

    uint8 frames [clips, T, 720, 1280, 3]  --preprocess (row f3)-->  fp32 [clips, 3, 384, 640]
      --trunk + heads + Fast NMS + temporal fusion + tracker (rows a1-a17)-->  tracked detections per clip
      --postprocess_ytbvis: crop / upsample / threshold / COCO RLE (row f1)-->  per-frame results
      --bbox2result_with_id + results2json_videoseg (row f2)-->  YouTube-VIS json

usage: python scripts/run_video_demo.py [--clips 2] [--frames 4] [--out /tmp/stmask_demo/results.json]
Synthetic video (seeded noise with a translation per frame) and seeded random weights: the point is the plumbing."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stmask_amd import eval_utils, output_utils, preprocess, synthetic  # noqa: E402
from stmask_amd.config import get_cfg  # noqa: E402
from stmask_amd.fuse import optimize_for_inference  # noqa: E402
from stmask_amd.model import STMask  # noqa: E402
from stmask_amd.pipeline import BatchedClipPipeline  # noqa: E402


def synthetic_video_u8(n_clips, n_frames, h=720, w=1280, seed=0):
    g = torch.Generator().manual_seed(seed)
    base = torch.randint(0, 256, (n_clips, h, w, 3), generator=g, dtype=torch.uint8)
    return torch.stack([torch.roll(base, shifts=(4 * t, 6 * t), dims=(1, 2)) for t in range(n_frames)], 1)


def run(n_clips=2, n_frames=4, config="STMask_plus_resnet50_config", out_file=None, dev="cuda"):
    net = STMask(get_cfg(config))
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=4.7)   # detection-seeding bias tuned to these frames' statistics (a handful of objects)
    net = net.to(dev)
    optimize_for_inference(net, planar=True)
    net = net.to(memory_format=torch.channels_last)
    net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    video = synthetic_video_u8(n_clips, n_frames).to(dev)
    pipe = BatchedClipPipeline(net, n_clips)
    classes = ["class_%d" % i for i in range(1, net.cfg.num_classes)]
    results = [[] for _ in range(n_clips)]
    with torch.no_grad():
        for t in range(n_frames):
            x, meta = preprocess.preprocess_eval_frames(video[:, t], idx=t)
            pipe.step(x.contiguous(memory_format=torch.channels_last), is_first=(t == 0))
            for c, det in enumerate(pipe.detections()):
                m = dict(meta, video_id=c)
                post = output_utils.postprocess_ytbvis({"detection": det}, m)
                results[c].append(eval_utils.bbox2result_with_id(post, m, classes))
    flat = [r for clip in results for r in clip]          # ordered by video, then frame, as results2json_videoseg assumes
    records = eval_utils.video_records(flat) if out_file is None else eval_utils.results2json_videoseg(flat, out_file)
    return records


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--config", default="STMask_plus_resnet50_config")
    ap.add_argument("--out", default="/tmp/stmask_demo/results.json")
    a = ap.parse_args()
    recs = run(a.clips, a.frames, a.config, a.out)
    n_seg = sum(1 for r in recs for s in r["segmentations"] if s is not None)
    print(json.dumps({"records": len(recs), "segmentations": n_seg, "out": a.out,
                      "videos": sorted({r["video_id"] for r in recs})}))
