#!/usr/bin/env python3
"""Throughput of the SURVEY 8(f) stages on either side of the hot path, at 720x1280 (BASELINE config 5's frame size) and 360x640:
  f3  pre-processing (eval.py:703-717: resize to the test scale, (x - mean) / std, pad to /32, HWC -> CHW) as ONE kernel on uint8 frames
      resident in HBM (stm_preprocess_u8_f32);
  f1  postprocess_ytbvis's mask leg (layers/output_utils.py:86-106: un-pad, bilinear upsample of the [n, h/4, w/4] soft masks to the frame
      size, > 0.5, COCO RLE) as device resize + threshold + run extraction (stm_mask_resize_rle_f32), then the D2H copy of the run lengths
      and the host-side 5-bit string packing -- against moving the full-resolution masks to the host (what the reference does).
HIP-event timing, median of 5 rounds x 10 launches.  usage: python scripts/bench_stages.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, output_utils      # noqa: E402

DEV = "cuda"


def timeit(fn, rounds=5, iters=10):
    fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters)
    res.sort()
    return res[len(res) // 2]


def blob_masks(n, mh, mw, seed=0):
    """soft masks with one smooth blob each (a few dozen runs per column-major RLE, like real instance masks)"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(mh).float(), torch.arange(mw).float(), indexing="ij")
    cy, cx = torch.rand(n, generator=g) * mh, torch.rand(n, generator=g) * mw
    ry, rx = 4 + torch.rand(n, generator=g) * mh / 3, 4 + torch.rand(n, generator=g) * mw / 3
    d = ((yy[None] - cy[:, None, None]) / ry[:, None, None]) ** 2 + ((xx[None] - cx[:, None, None]) / rx[:, None, None]) ** 2
    return torch.sigmoid(4.0 * (1.0 - d))


print("# f3: pre-processing, uint8 HWC frames resident in HBM -> fp32 CHW normalised, padded to /32 (one kernel)")
for (n, H0, W0, size, tag) in [(32, 720, 1280, (1280, 720), "720x1280 -> 736x1280 (config 5)"), (32, 720, 1280, (640, 360), "720x1280 -> 384x640 (resize)"),
                               (32, 360, 640, (640, 360), "360x640 -> 384x640")]:
    img = torch.randint(0, 256, (n, H0, W0, 3), dtype=torch.uint8, device=DEV)
    out = ops.preprocess_frames(img, size=size)
    ms = timeit(lambda: ops.preprocess_frames(img, size=size))
    nbytes = img.numel() + out.numel() * 4
    print(f"f3 {tag:34s} {n:3d} frames  {ms * 1e3:8.1f} us  = {n / ms * 1e3:9.0f} frames/s   {nbytes / ms / 1e6:7.1f} GB/s (uint8 in + fp32 out)", flush=True)

print("# f1: soft masks [n, mh, mw] -> un-pad, bilinear upsample to the frame, > 0.5, COCO RLE run lengths (device), + D2H of the runs + string packing (host)")
for (n, mh, mw, crop_h, crop_w, oh, ow, tag) in [(100, 184, 320, 180, 320, 720, 1280, "720x1280, 100 masks"), (1000, 184, 320, 180, 320, 720, 1280, "720x1280, 1000 masks"),
                                                 (100, 96, 160, 90, 160, 360, 640, "360x640, 100 masks"), (1000, 96, 160, 90, 160, 360, 640, "360x640, 1000 masks")]:
    m = blob_masks(n, mh, mw).to(DEV)
    ms = timeit(lambda: ops.mask_resize_rle(m, crop_h, crop_w, oh, ow))
    px = n * oh * ow
    t0 = time.perf_counter()
    for _ in range(3):
        rles = output_utils.encode_masks(m, crop_h, crop_w, oh, ow)
    host_ms = (time.perf_counter() - t0) / 3 * 1e3
    runs = sum(len(r["counts"]) for r in rles)
    # the reference's way: upsample on the device, then every full-resolution binary mask crosses PCIe (output_utils.py:101-106)
    up = torch.nn.functional.interpolate(m[None, :, :crop_h, :crop_w], (oh, ow), mode="bilinear", align_corners=False)[0].gt_(0.5)
    t0 = time.perf_counter()
    for _ in range(3):
        host = up.to(torch.uint8).cpu()
    torch.cuda.synchronize()
    ref_ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"f1 {tag:22s} device kernel {ms * 1e3:8.1f} us = {px / ms / 1e6:7.1f} Gpixel/s ({n / ms * 1e3:8.0f} masks/s);  end to end with D2H + strings "
          f"{host_ms:7.2f} ms ({runs / n:5.0f} B of RLE per mask);  D2H of the full-resolution masks alone: {ref_ms:7.2f} ms ({n * oh * ow / 1e6:.0f} MB)", flush=True)
