#!/usr/bin/env python3
"""Timing ablations of conv_planar_kernel in the fp16x2 (or bf16x3) format on the large layers of the graph: which part of
the K loop bounds it.  STM_CONV_DEBUG bits (results are wrong with any of them): 1 no LDS-DMA in the loop, 2 no barrier,
4 no MFMA.  usage: python scripts/ablate_planar.py [fmt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LAYERS = [("proto 3x3 96x160", 96, 160, 256, 256, 3), ("tower 3x3 P3 256->1024", 48, 80, 256, 1024, 3), ("3x3 48x80 256->256", 48, 80, 256, 256, 3),
          ("1x1 96x160 64->256", 96, 160, 64, 256, 1), ("1x1 48x80 128->512", 48, 80, 128, 512, 1)]


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


for name, H, W, C, O, k in LAYERS:
    x = torch.randn(8, H, W, C, device="cuda")
    w = torch.randn(O, C, k, k, device="cuda") * (C * k * k) ** -0.5
    b = torch.randn(O, device="cuda")
    if fmt == 1:
        pk, osc = ops.conv_pack_weights(w, fmt=1)
    else:
        pk, osc = ops.conv_pack_weights(w, 3), 1.0
    xp = ops.split_planes(x, fmt)
    gf = 2.0 * 8 * H * W * C * O * k * k / 1e9
    row = []
    for dbg in (0, 4, 1, 5, 2, 6, 3):
        os.environ["STM_CONV_DEBUG"] = str(dbg)
        us = timeit(lambda: ops.conv2d_planar(xp, pk, (O, C, k, k), (8, H, W), b, None, padding=k // 2, relu=True, out="planes", fmt=fmt,
                                              out_scale=osc))
        row.append("dbg%d %7.1f us" % (dbg, us))
    os.environ["STM_CONV_DEBUG"] = "0"
    print("%-26s %6.1f GF  " % (name, gf) + "  ".join(row) + "   (full: %.0f TF)" % (gf / float(row[0].split()[1]) * 1e-3 * 1e3))
