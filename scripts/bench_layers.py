#!/usr/bin/env python3
"""Per-layer microbenchmark of the planar convolution on the layer shapes of the benchmark graph (R50-DCN-FPN, 384x640, batch 32):
true kernel durations from HIP events around back-to-back launches, rotating over several input / output buffers so that
the working set exceeds the 256-MB Infinity Cache (cache-resident figures flatter the HBM-bound layers by 1.5-2x).

    python scripts/bench_layers.py [--fmt 1] [--set hbm|mfma|all] [--batch 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, planar  # noqa: E402
from stmask_amd.planar import PlanarConv  # noqa: E402

# (name, H, W, C, O, k, stride, residual, out)   at batch B
HBM = [("l1 conv3 64->256 +res", 96, 160, 64, 256, 1, 1, True, "planes"),
       ("l1 conv1 256->64", 96, 160, 256, 64, 1, 1, False, "planes"),
       ("l1 conv2 64->64 3x3", 96, 160, 64, 64, 3, 1, False, "planes"),
       ("l2 conv3 128->512 +res", 48, 80, 128, 512, 1, 1, True, "planes"),
       ("l2 conv1 512->128", 48, 80, 512, 128, 1, 1, False, "both"),
       ("l2 ds 256->512 s2", 96, 160, 256, 512, 1, 2, False, "planes"),
       ("l3 conv3 256->1024 +res", 24, 40, 256, 1024, 1, 1, True, "planes"),
       ("l3 conv1 1024->256", 24, 40, 1024, 256, 1, 1, False, "both"),
       ("l4 conv3 512->2048 +res", 12, 20, 512, 2048, 1, 1, True, "planes"),
       ("l4 conv1 2048->512", 12, 20, 2048, 512, 1, 1, False, "both")]
MFMA = [("proto 256->256 3x3 @96x160", 96, 160, 256, 256, 3, 1, False, "planes"),
        ("fpn/proto 256->256 3x3 @48x80", 48, 80, 256, 256, 3, 1, False, "planes"),
        ("tower1 256->1024 3x3 @levels(48x80 proxy x1.33)", 48, 107, 256, 1024, 3, 1, False, "planes"),
        ("dcn gemm l2 1152->128", 48, 80, 1152, 128, 1, 1, False, "planes"),
        ("dcn gemm l3 2304->256", 24, 40, 2304, 256, 1, 1, False, "planes"),
        ("temporal conv3 512->1024 3x3 (2900 rois)", 7, 7 * 91, 512, 1024, 3, 1, False, "f32")]


EXP = [("64->256 +res planes (baseline)", 96, 160, 64, 256, 1, 1, True, "planes"),
       ("64->256 no res", 96, 160, 64, 256, 1, 1, False, "planes"),
       ("64->256 +res f32 out", 96, 160, 64, 256, 1, 1, True, "f32"),
       ("64->64 1x1 (one n-tile)", 96, 160, 64, 64, 1, 1, False, "planes"),
       ("256->256 1x1", 96, 160, 256, 256, 1, 1, False, "planes"),
       ("64->256 +res tile128", 96, 160, 64, 256, 1, 1, True, "planes", 128),
       ("128->512 +res tile128", 48, 80, 128, 512, 1, 1, True, "planes", 128),
       ("256->1024 +res tile128", 24, 40, 256, 1024, 1, 1, True, "planes", 128)]


# single-round grids at batch 32: 240 tiles of 256 pixels on 256 CUs (--set mg: run under STM_CONV_MG=1 / 2 to compare the pixel-tile sizes)
MG = [("l3 conv1 1024->256 1x1 @24x40", 24, 40, 1024, 256, 1, 1, False, "planes"),
      ("l3->l4 1024->512 1x1 @24x40", 24, 40, 1024, 512, 1, 1, False, "planes"),
      ("fpn 256->256 3x3 @24x40", 24, 40, 256, 256, 3, 1, False, "planes"),
      ("dcn gemm l3 2304->256 @24x40", 24, 40, 2304, 256, 1, 1, False, "planes"),
      ("l4 conv1 2048->512 1x1 @12x20", 12, 20, 2048, 512, 1, 1, False, "planes"),
      ("dcn gemm l4 4608->512 @12x20", 12, 20, 4608, 512, 1, 1, False, "planes")]


def run(name, B, H, W, C, O, k, s, has_res, out, tile_n=None, fmt=1, reps=20, nbuf=4):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn(O, C, k, k, device=dev, generator=g) * (C * k * k) ** -0.5
    b = torch.randn(O, device=dev, generator=g)
    conv = PlanarConv(w, b, s, k // 2, relu=True, fmt=fmt, tile_n=tile_n)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * Ho * Wo
    xs = [ops.split_planes(torch.randn(B, H, W, C, device=dev, generator=g), fmt) for _ in range(nbuf)]
    rs = [ops.split_planes(torch.randn(M, O, device=dev, generator=g), fmt) for _ in range(nbuf)] if has_res else [None] * nbuf
    NP, dt = ops.plane_layout(fmt)
    ops_ = [torch.empty(NP, -(-O // 32), M, 32, device=dev, dtype=dt) for _ in range(nbuf)] if out in ("planes", "both") else [None] * nbuf
    of = [torch.empty(M, O, device=dev) for _ in range(nbuf)] if out in ("f32", "both") else [None] * nbuf
    call = lambda i: conv(xs[i % nbuf], ("img", B, H, W), out=out, out_planes=ops_[i % nbuf], out_f32=of[i % nbuf], residual=rs[i % nbuf])
    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * M * O * C * k * k
    pb = 2 * NP
    byts = B * H * W * C * pb + M * O * (pb if out != "f32" else 0) + M * O * (4 if out != "planes" else 0) + (M * O * pb if has_res else 0)
    print(f"{name:52s} M={M:7d} tile={conv.pick_tile(M):3d} {us:8.1f} us  {flops / us / 1e6:7.1f} TF  {byts / us / 1e6:6.2f} TB/s  ({byts / 1e6:.0f} MB)", flush=True)
    return us


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fmt", type=int, default=1)
    ap.add_argument("--set", default="all")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    planar.set_format(a.fmt if a.fmt != 2 else 1, backbone_fmt=2 if a.fmt == 2 else None)
    print(torch.cuda.get_device_name(0), "fmt", a.fmt, "batch", a.batch, flush=True)
    tot = 0.0
    for name, *shape in (HBM if a.set in ("hbm", "all") else []) + (MFMA if a.set in ("mfma", "all") else []) + (EXP if a.set == "exp" else []) + (MG if a.set == "mg" else []) + (MFMA[:1] if a.set == "proto" else []) + (EXP[:1] if a.set == "one" else []):
        tot += run(name, a.batch, *shape, fmt=a.fmt)
    print(f"sum {tot:.0f} us")
