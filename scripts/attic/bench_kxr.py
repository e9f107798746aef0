#!/usr/bin/env python3
"""A/B of the narrow-output layers: stm_conv2d_planar_kxr_f32 (csrc/conv_kxr.hip) against the general planar kernel's 128 x 64 tiles,
at the benchmark's shapes (batch 32, 384x640): the head's output layers (3 groups of 41 / 5 / 32 channels, kernels 3x3 / 3x5 / 5x3,
five levels), layer1's 64 -> 64 3x3, the stride-1 DCN offset convolutions.  usage: python scripts/bench_kxr.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops                      # noqa: E402
from stmask_amd.planar import PlanarConv        # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LEVELS = [(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, kh, kw, groups, cg, real, C, shape, relu=False):
    torch.manual_seed(0)
    w = torch.randn(groups * cg, C, kh, kw, device=DEV) * (C * kh * kw) ** -0.5
    b = torch.randn(groups * cg, device=DEV)
    M = sum(B * h * w_ for h, w_ in shape[2]) if shape[0] == "levels" else shape[1] * shape[2] * shape[3]
    xp = (torch.randn(2, groups * C // 32, M, 32, device=DEV) * 0.5).half()
    res = {}
    for kxr in (False, True):
        conv = PlanarConv(w, b, 1, ((kh - 1) // 2, (kw - 1) // 2), relu=relu, groups=groups, group_cout=list(real), tile_n=64, fmt=1)
        conv.kxr = kxr and ops.conv_kxr_supported(groups * cg, C, kh, kw, 1, ((kh - 1) // 2, (kw - 1) // 2), groups, list(real), 1)
        conv.kxr_min_pixels = 0
        out = torch.empty(M, groups * cg, device=DEV)
        res[kxr] = timeit(lambda: conv(xp, shape, out="f32", out_f32=out))
        res[("y", kxr)] = out.clone()
    gf = 2.0 * M * sum(real) * C * kh * kw / 1e9
    d = max((res[("y", True)][:, g * cg:g * cg + real[g]] - res[("y", False)][:, g * cg:g * cg + real[g]]).abs().max().item() for g in range(groups))
    print(f"{name:34s} M={M:7d}  general {res[False]:8.1f} us ({gf / res[False] * 1e3:6.1f} TF)   kxr {res[True]:8.1f} us ({gf / res[True] * 1e3:6.1f} TF)   "
          f"x{res[False] / res[True]:.2f}   max diff {d:.2e}", flush=True)


if __name__ == "__main__":
    lv = ("levels", B, LEVELS)
    case("head out 3x3 (41,5,32)", 3, 3, 3, 64, (41, 5, 32), 256, lv)
    case("head out 3x5 (41,5,32)", 3, 5, 3, 64, (41, 5, 32), 256, lv)
    case("head out 5x3 (41,5,32)", 5, 3, 3, 64, (41, 5, 32), 256, lv)
    case("FCB trailing conv 3x3 -> 41", 3, 3, 1, 64, (41,), 256, lv)
    case("layer1 64->64 3x3 @96x160", 3, 3, 1, 64, (64,), 64, ("img", B, 96, 160), relu=True)
    case("DCN offset 128->27 @48x80", 3, 3, 1, 32, (27,), 128, ("img", B, 48, 80))
    case("DCN offset 256->27 @24x40", 3, 3, 1, 32, (27,), 256, ("img", B, 24, 40))
    case("DCN offset 512->27 @12x20", 3, 3, 1, 32, (27,), 512, ("img", B, 12, 20))
