#!/usr/bin/env python3
"""A/B of the planar conv K-loop variants on the big layers (fp16 format): STM_CONV_RING=2|3 is read once per process, so
this script is run once per setting.  usage: STM_CONV_RING=3 python scripts/ab_ring.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
LAYERS = [("proto 3x3 96x160 256->256", 96, 160, 256, 256, 3), ("tower 3x3 P3 256->1024", 48, 80, 256, 1024, 3), ("3x3 48x80 256->256", 48, 80, 256, 256, 3),
          ("3x3 24x40 256->256 (mg1)", 24, 40, 256, 256, 3), ("1x1 96x160 64->256", 96, 160, 64, 256, 1), ("1x1 48x80 128->512", 48, 80, 128, 512, 1),
          ("1x1 24x40 256->1024", 24, 40, 256, 1024, 1), ("tnet 3x3 ~40k 512->1024", 41, 123, 512, 1024, 3)]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("STM_CONV_RING =", os.environ.get("STM_CONV_RING", "(default 3)"), " STM_CONV_ABL =", os.environ.get("STM_CONV_ABL", "0"))
if os.environ.get("STM_CONV_ABL"):
    LAYERS = LAYERS[:3]
for name, H, W, C, O, k in LAYERS:
    x = torch.randn(8, H, W, C, device="cuda")
    w = torch.randn(O, C, k, k, device="cuda") * (C * k * k) ** -0.5
    b = torch.randn(O, device="cuda")
    pk, osc = ops.conv_pack_weights(w, fmt=1)
    xp = ops.split_planes(x, 1)
    gf = 2.0 * 8 * H * W * C * O * k * k / 1e9
    us = timeit(lambda: ops.conv2d_planar(xp, pk, (O, C, k, k), (8, H, W), b, None, padding=k // 2, relu=True, out="planes", fmt=1, out_scale=osc))
    print("%-28s %6.1f GF %8.1f us %7.1f TF" % (name, gf, us, gf / us * 1e3))
