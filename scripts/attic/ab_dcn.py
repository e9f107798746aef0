#!/usr/bin/env python3
"""A/B of the planar deformable sampler's launch options (STM_DCN_XCD, STM_DCN_NT are read per launch) on the R50 DCN layer
shapes.  usage: python scripts/ab_dcn.py [batch] [fmt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fmt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LAYERS = [("layer2 s1 128ch 48x80", 128, 48, 80, 1), ("layer2 s2 128ch 96x160", 128, 96, 160, 2), ("layer3 256ch 24x40", 256, 24, 40, 1),
          ("layer4 512ch 12x20", 512, 12, 20, 1)]


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


for name, C, H, W, s in LAYERS:
    x = torch.randn(B, H, W, C, device="cuda")
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    om = torch.randn(B * Ho * Wo, 27, device="cuda") * 1.5
    nbytes = 4 * B * C * H * W + 4 * 27 * B * Ho * Wo + (4 if fmt == 1 else 6) * 9 * C * B * Ho * Wo
    row = []
    for xcd, nt in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        os.environ["STM_DCN_XCD"], os.environ["STM_DCN_NT"] = str(xcd), str(nt)
        us = timeit(lambda: ops.dcn_sample_planar(x, om, s, 1, 1, fmt=fmt))
        row.append("xcd%d nt%d %7.1f us %5.2f TB/s" % (xcd, nt, us, nbytes / us / 1e6))
    print("%-26s %6.1f MB  " % (name, nbytes / 1e6) + " | ".join(row))
