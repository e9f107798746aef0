#!/usr/bin/env python3
"""Launch-heuristic sweep for the small-grid layers (layer3 / layer4 / head shapes at 8 clips): every configuration is captured in
a HIP graph of 20 back-to-back launches (rotating buffers) and timed by replay, so host launch cost is out of the figure.

    python scripts/sweep_small_m.py [--batch 8]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import _lib, ops  # noqa: E402
from stmask_amd.planar import PlanarConv  # noqa: E402

# (name, H, W, C, O, k, has_res, out)
SHAPES = [("l3 conv1 1024->256", 24, 40, 1024, 256, 1, False, "both"),
          ("l3 conv3 256->1024 +res", 24, 40, 256, 1024, 1, True, "planes"),
          ("l3 dcn gemm 2304->256", 24, 40, 2304, 256, 1, False, "planes"),
          ("fpn 256->256 3x3 @24x40", 24, 40, 256, 256, 3, False, "planes"),
          ("l4 conv1 2048->512", 12, 20, 2048, 512, 1, False, "both"),
          ("l4 dcn gemm 4608->512", 12, 20, 4608, 512, 1, False, "planes"),
          ("l2 conv1 512->128", 48, 80, 512, 128, 1, False, "both"),
          ("head 256->256 3x3 (M=5115 at 1 clip)", 55, 93, 256, 256, 3, False, "planes"),
          ("head 256->128 3x3 (M=5115 at 1 clip)", 55, 93, 256, 128, 3, False, "planes"),
          ("proto 256->256 3x3 @48x80", 48, 80, 256, 256, 3, False, "planes"),
          ("l2 conv2 128->128 3x3 @48x80", 48, 80, 128, 128, 3, False, "planes"),
          ("l1 conv2 64->64 3x3 @96x160", 96, 160, 64, 64, 3, False, "planes"),
          ("l1 conv1 256->64 @96x160", 96, 160, 256, 64, 1, False, "planes"),
          ("l1 conv3 64->256 +res @96x160", 96, 160, 64, 256, 1, True, "planes"),
          ("l2 conv3 128->512 +res @48x80", 48, 80, 128, 512, 1, True, "planes"),
          ("l2 dcn gemm 1152->128 @48x80", 48, 80, 1152, 128, 1, False, "planes"),
          ("proto 256->256 3x3 @96x160", 96, 160, 256, 256, 3, False, "planes"),
          ("tower 256->1024 3x3 (levels proxy 55x93)", 55, 93, 256, 1024, 3, False, "planes"),
          ("l4 conv3 512->2048 +res", 12, 20, 512, 2048, 1, True, "planes")]


def set_env(env):
    for k in list(os.environ):
        if k.startswith("STM_CONV_"):
            del os.environ[k]
    os.environ.update(env)
    _lib.lib().stm_debug_reload_tunables()


def timed(conv, B, H, W, C, O, has_res, out, fmt=1, nbuf=4, reps=20):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    M = B * H * W
    xs = [ops.split_planes(torch.randn(B, H, W, C, device=dev, generator=g), fmt) for _ in range(nbuf)]
    rs = [ops.split_planes(torch.randn(M, O, device=dev, generator=g), fmt) for _ in range(nbuf)] if has_res else [None] * nbuf
    call = lambda i: conv(xs[i % nbuf], ("img", B, H, W), out=out, residual=rs[i % nbuf])
    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(reps):
            call(i)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--shapes", default="", help="comma-separated substrings selecting shapes")
    a = ap.parse_args()
    if a.shapes:
        SHAPES = [s_ for s_ in SHAPES if any(t in s_[0] for t in a.shapes.split(","))]
    dev = "cuda"
    configs = [("default", {}, None)]
    for sk in (1, 2, 3, 4, 6, 8):
        for tile in (64, 128):
            configs.append((f"sk{sk} t{tile}", {"STM_CONV_SPLITK": str(sk)}, tile))
    configs += [("sk3 t64 ring", {"STM_CONV_SPLITK": "3", "STM_CONV_RING64": "4"}, 64), ("sk2 t64 ring", {"STM_CONV_SPLITK": "2", "STM_CONV_RING64": "4"}, 64),
                ("sk1 t64 ring", {"STM_CONV_SPLITK": "1", "STM_CONV_RING64": "4"}, 64), ("sk4 t64 ring", {"STM_CONV_SPLITK": "4", "STM_CONV_RING64": "4"}, 64),
                ("sk1 t64 2buf", {"STM_CONV_SPLITK": "1", "STM_CONV_RING64": "2"}, 64), ("sk1 t128 mg1", {"STM_CONV_SPLITK": "1", "STM_CONV_MG": "1"}, 128),
                ("sk2 t128 mg1", {"STM_CONV_SPLITK": "2", "STM_CONV_MG": "1"}, 128)]
    print(torch.cuda.get_device_name(0), "batch", a.batch)
    for name, H, W, C, O, k, has_res, out in SHAPES:
        g = torch.Generator(device=dev).manual_seed(1)
        w = torch.randn(O, C, k, k, device=dev, generator=g) * (C * k * k) ** -0.5
        b = torch.randn(O, device=dev, generator=g)
        res = []
        for cname, env, tile in configs:
            set_env(env)
            conv = PlanarConv(w, b, 1, k // 2, relu=True, fmt=1, tile_n=tile)
            us = timed(conv, a.batch, H, W, C, O, has_res, out)
            res.append((us, cname))
        set_env({})
        conv = PlanarConv(w, b, 1, k // 2, relu=True, fmt=1, tile_n=None)
        res[0] = (timed(conv, a.batch, H, W, C, O, has_res, out), "default")      # again, warm: the first figure includes the clock ramp
        base = res[0][0]
        best = sorted(res)[:5]
        print(f"{name:28s} M={a.batch * H * W:6d} default {base:6.1f} us | best: " + ", ".join(f"{c} {u:.1f}" for u, c in best), flush=True)
